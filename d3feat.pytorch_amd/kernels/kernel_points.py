"""Kernel-point dispositions for KPConv.

Mirror of the reference's ``kernels.kernel_points`` host code (kernels/kernel_points.py): ``load_kernels(radius,
num_kpoints, dimension, fixed)`` (:400-482) returns a float32 ``[K, dimension]`` disposition scaled by ``radius``,
with the run-time randomisation the reference applies at model construction (one ``np.random.rand`` for the rotation
about z, N(0, 0.01) jitter; both from the global NumPy RNG, :446-480).  With the same global seed the result equals the
reference's, which makes a random-init model comparable seed for seed.  KPConv stores it as the non-trainable
``kernel_points`` parameter, so a reference checkpoint overrides it on ``load_state_dict`` anyway.

The base disposition (before rotation and jitter) comes from, in this order,
  1. the in-memory cache,
  2. a data file ``dispositions/k_{K:03d}_{fixed}_{dim}D.npy`` next to this module -- K=15/center/3D is the table the
     reference ships as kernels/dispositions/k_015_center_3D.ply (data, the one its configs use),
  3. the optimiser below: ``kernel_point_optimization`` (:258-396; 100 candidate kernels, the one with the smallest
     final gradient norm is kept, :426-439) for K <= 30, ``spherical_lloyd`` (:78-254) above that (:411-412).
Both optimisers draw from the global NumPy RNG in the reference's order, so they too are reproducible seed for seed
(tests/golden/kernel_points.npz holds reference runs).  This is host-side set-up code run once per layer type.
"""
import os

import numpy as np

_CACHE = {}
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dispositions')


def _pin(points, fixed, r0=1.0):
    """Fixed points of a disposition(s) [..., K, dim]: the centre, or centre + two on the vertical axis."""
    if fixed == 'center':
        points[..., 0, :] *= 0
    if fixed == 'verticals':
        points[..., :3, :] *= 0
        points[..., 1, -1] += 2 * r0 / 3
        points[..., 2, -1] -= 2 * r0 / 3


def kernel_point_optimization(radius, num_points, num_kernels=1, dimension=3, fixed='center', ratio=0.66):
    """Repulsion optimisation of ``num_kernels`` independent dispositions (reference kernel_points.py:258-396).

    Energy per kernel: sum over pairs 1/d (points repel) + 5 |x|^2 (pull to the centre).  Steepest descent with a step
    of min(lr * |grad|, 0.05) per point, lr = 1e-2 decaying by 0.9995 per iteration, at most 10000 iterations, stopped
    when no free point's gradient norm changed by 1e-5.  Returns (points [num_kernels, K, dim] * radius, history of
    max gradient norms [10000, num_kernels]); the shell is rescaled so that the mean radius of points 1.. is ``ratio``.
    """
    r0, lr, decay, tol, clip, max_iter = 1.0, 1e-2, 0.9995, 1e-5, 0.05, 10000
    want = num_kernels * num_points
    # rejection sampling inside the ball of squared radius 0.5 (the first draw is kept whole, :296-302)
    pts = np.random.rand(want - 1, dimension) * 2 * r0 - r0
    while pts.shape[0] < want:
        more = np.random.rand(want - 1, dimension) * 2 * r0 - r0
        pts = np.vstack((pts, more))
        pts = pts[np.sum(np.power(pts, 2), axis=1) < 0.5 * r0 * r0, :]
    pts = pts[:want, :].reshape((num_kernels, num_points, -1))
    _pin(pts, fixed, r0)
    first_free = {'center': 1, 'verticals': 3}.get(fixed, 0)

    history = np.zeros((max_iter, num_kernels))
    previous = np.zeros((num_kernels, num_points))
    for it in range(max_iter):
        a = np.expand_dims(pts, axis=2)
        b = np.expand_dims(pts, axis=1)
        d2 = np.sum(np.power(a - b, 2), axis=-1)
        grad = np.sum((a - b) / (np.power(np.expand_dims(d2, -1), 3 / 2) + 1e-6), axis=1) + 10 * pts
        if fixed == 'verticals':
            grad[:, 1:3, :-1] = 0
        norms = np.sqrt(np.sum(np.power(grad, 2), axis=-1))
        history[it, :] = np.max(norms, axis=1)
        if np.max(np.abs(previous[:, first_free:] - norms[:, first_free:])) < tol:
            break
        previous = norms
        step = np.minimum(lr * norms, clip)
        if fixed in ('center', 'verticals'):
            step[:, 0] = 0
        pts -= np.expand_dims(step, -1) * grad / np.expand_dims(norms + 1e-6, -1)
        lr *= decay
    shell = np.sqrt(np.sum(np.power(pts, 2), axis=-1))
    pts *= ratio / np.mean(shell[:, 1:])
    return pts * radius, history


def spherical_lloyd(radius, num_cells, dimension=3, fixed='center', approx_n=5000, max_iter=500, momentum=0.9):
    """Monte-Carlo Lloyd relaxation of ``num_cells`` points in the unit ball (reference kernel_points.py:78-254,
    approximation='monte-carlo'): every iteration draws ``approx_n`` points of the cube, keeps those in the ball,
    assigns them to their nearest kernel point and moves each kernel point 1 - momentum of the way to its cell's
    centroid.  Returns points [num_cells, dim] * radius."""
    r0 = 1.0
    pts = np.zeros((0, dimension))
    while pts.shape[0] < num_cells:       # start inside the outer shell 0.9 < |x| < 1 (:108-114)
        pts = np.vstack((pts, np.random.rand(num_cells, dimension) * 2 * r0 - r0))
        d2 = np.sum(np.power(pts, 2), axis=1)
        pts = pts[np.logical_and(d2 < r0 ** 2, (0.9 * r0) ** 2 < d2), :]
    pts = pts[:num_cells, :].reshape((num_cells, -1))
    _pin(pts, fixed, r0)
    for _ in range(max_iter):
        x = np.random.rand(approx_n, dimension) * 2 * r0 - r0
        x = x[np.sum(np.power(x, 2), axis=1) < r0 * r0, :]
        cell = np.argmin(np.sum(np.square(np.expand_dims(x, 1) - pts), axis=2), axis=1)
        centers = pts.copy()
        for c in range(num_cells):
            member = cell == c
            n = np.sum(member.astype(np.int32))
            if n > 0:
                centers[c] = np.sum(x[member, :], axis=0) / n
        pts += (1 - momentum) * (centers - pts)
        if fixed == 'center':
            pts[0, :] *= 0
        if fixed == 'verticals':
            pts[0, :] *= 0
            pts[:3, :-1] *= 0
    return pts * radius


def _name(num_kpoints, dimension, fixed):
    return 'k_{:03d}_{:s}_{:d}D.npy'.format(num_kpoints, fixed, dimension)


def base_disposition(num_kpoints, dimension=3, fixed='center', lloyd=False):
    """Unit-radius disposition [K, dim] (float64) before the per-layer rotation and jitter."""
    key = (int(num_kpoints), int(dimension), str(fixed))
    if key not in _CACHE:
        path = os.path.join(_DIR, _name(*key[:1], key[1], key[2]))
        if os.path.exists(path):
            _CACHE[key] = np.load(path)
        elif lloyd or num_kpoints > 30:
            _CACHE[key] = spherical_lloyd(1.0, num_kpoints, dimension=dimension, fixed=fixed)
        else:
            cand, hist = kernel_point_optimization(1.0, num_kpoints, num_kernels=100, dimension=dimension, fixed=fixed)
            _CACHE[key] = cand[np.argmin(hist[-1, :]), :, :]     # the reference's pick (:437): last history row
    return _CACHE[key].copy()


def load_kernels(radius, num_kpoints, dimension, fixed, lloyd=False):
    """float32 [K, dimension] kernel points for one KPConv layer (reference kernel_points.py:400-482)."""
    if dimension not in (2, 3):
        raise ValueError('Unsupported dimension of kernel : ' + str(dimension))
    kp = base_disposition(num_kpoints, dimension, fixed, lloyd)
    R = np.eye(dimension)
    theta = np.random.rand() * 2 * np.pi
    c, s = np.cos(theta), np.sin(theta)
    if fixed != 'vertical':               # the reference tests 'vertical' (never a value of `fixed`), :450-455
        R = (np.array([[c, -s], [s, c]], dtype=np.float32) if dimension == 2 else
             np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32))
    elif dimension == 3:                  # random axis + angle (:457-470), Rodrigues form
        phi = (np.random.rand() - 0.5) * np.pi
        u = np.array([np.cos(theta) * np.cos(phi), np.sin(theta) * np.cos(phi), np.sin(phi)])
        alpha = np.random.rand() * 2 * np.pi
        ux = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
        R = (np.cos(alpha) * np.eye(3) + np.sin(alpha) * ux + (1 - np.cos(alpha)) * np.outer(u, u)).astype(np.float32)
    kp = kp + np.random.normal(scale=0.01, size=kp.shape)
    kp = radius * kp
    kp = np.matmul(kp, R)
    return kp.astype(np.float32)
