/* d3feat_hip.h -- C ABI of libd3feat_hip.so, the MI355X (gfx950) implementation of the D3Feat hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - all tensors are dense row-major; point clouds are [N,3] float32, features [N,C] float32,
 *     neighbor tables [Nq,H] int32 with the reference's shadow convention: entry == Ns (number of
 *     support rows) means "no neighbor" (reference neighbors.cpp:324, blocks.py:277,356);
 *   - batch (stack) lengths are int32 device arrays of B entries, like the reference's q_batches /
 *     s_batches (cpp_neighbors/wrapper.cpp:85-86);
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous and stream-ordered, never
 *     synchronise, never allocate; scratch comes from the caller (`ws`, size from *_ws_bytes);
 *   - return value: 0 on success, negative D3F_E* on a host-detectable argument error.  Device-detected
 *     conditions (candidate overflow, cell-range overflow) are OR-ed into the int32 status word the
 *     caller supplies; the Python layer maps both to RuntimeError, the reference's only error type
 *     (cpp_neighbors/wrapper.cpp:77,95,...).
 *
 * Each entry point names the reference interface it replaces.
 */
#ifndef D3FEAT_HIP_H_
#define D3FEAT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3F_OK 0
#define D3F_EINVAL (-1)     /* bad argument (shape / null / unsupported size) */
#define D3F_EWORKSPACE (-2) /* workspace too small */
#define D3F_ELAUNCH (-3)    /* HIP launch failure */

/* bits of the device status word */
#define D3F_ST_CAND_OVERFLOW 1  /* a query had more in-radius candidates than the kernel can rank */
#define D3F_ST_CELL_RANGE 2     /* a point fell outside the +-32767-cell addressable grid */
#define D3F_ST_TABLE_FULL 4     /* voxel hash table full (workspace sized for fewer points) */
#define D3F_ST_CAPACITY 8       /* an output needed more rows than the caller's capacity */
#define D3F_ST_WIDE_OVERFLOW 16 /* a point has more in-radius neighbors than the wide (reverse) table holds */
#define D3F_ST_NO_NEAREST 32    /* d3f_radius_query_prefix with a nearest bound: a query has no support within the bound */

const char* d3f_version(void);
int d3f_device_arch_ok(void); /* 1 if the current HIP device is gfx950, 0 otherwise, <0 = -(hipError_t) */
int d3f_device_arch_name(char* out, int n); /* gcnArchName of the current device */
void d3f_debug_set_flags(int flags);      /* profiling aid: ablation switches of the fused KPConv kernels (0 = off) */
/* measurement aid (profiles/phase_clock.py): `counters` = device uint64 buffer (or NULL to switch it off): [0] = waves
 * recorded (cleared by the caller), [1] = record capacity, record r at [8 + 8 r .. +5] = shader cycles one wave of the
 * fused KPConv forward / gather-form grad-input kernel spent in each of its phases (see the kernels). */
void d3f_debug_set_phase_clock(void* counters);
/* measurement aid (bench.py roofline leg): HIP events on the launch stream around every launch of ONE kernel
 * (which = 1 fused KPConv forward kernel, 2 scatter-form grad-input kernel, 3 gather-form grad-input kernel, 4 the
 * A^T B weight-gradient kernels launched one problem at a time, 5 / 6 the forward / transposed KPConv aggregation
 * kernels, 7 the GROUPED A^T B weight-gradient launches (d3f_linear_grad_weight_group: one record per group, both
 * launches); or, negative, minus a bit mask of several: -(1 | 2 | ... | 64)) between begin and end.  end -- after the
 * caller synchronised the device -- returns the number of launches seen and fills ms_out[i] and
 * shapes_out[6*i .. 6*i+5] = {Nq, Ns, H, Cin, Cout, K | which << 8} for the first `cap` of them (which = 7:
 * {problems, sum of 2 R M N in MiFLOP, sum of 4 R (M + N) + 4 M N in KiB, workgroups, 0, 7 << 8}). */
int d3f_debug_kernel_timing_begin(int which, int max_launches);
int d3f_debug_kernel_timing_end(float* ms_out, int32_t* shapes_out, int cap);

/* Tunables of the library -- ONE struct instead of environment variables: the library itself never reads the
 * environment.  d3f_get_tunables fills the current values (defaults at load), d3f_set_tunables replaces them all
 * (process-wide; set them before the launches they concern, not concurrently with them).  The experiment scripts under
 * profiles/ and the tests are the only callers; 0 = "the built-in choice" for every field. */
typedef struct d3f_tunables {
  int32_t atb_task_us;        /* grouped A^T B: modelled duration of one workgroup's task in us (0 = 40) */
  int32_t atb_form;           /* one-problem weight gradients: 0 = by size, 1 = always the first (direct-load) form,
                               * 2 = always the grouped (LDS-DMA ring) kernels */
  int32_t atb_first_form_wgs; /* first form: workgroups along the reduction (0 = by shape) */
  int32_t match_wgs;          /* d3f_mutual_nn: target number of workgroups (0 = 2048) */
  int32_t agg_through_lds;    /* general-path KPConv aggregation: 1 = stage the tile through LDS (experiment) */
  int32_t atb_pipe;           /* grouped A^T B task body: 0 = software-pipelined, 1 = the plain one, n >= 2 = pipelined only for
                               * tiles of at least n 16x16 accumulators (A/B measurements) */
  int32_t xw_rows;            /* d3f_gemm_epilogue: 0 = by shape, 2 / 4 = 32- / 64-row output blocks (A/B measurements) */
  int32_t xw_split;           /* d3f_gemm_epilogue: 0 = by shape, 1 = never split the reduction, 8 q = 8 q partitions */
  int32_t rowgemm_wide;       /* row-streaming unary kernels, 4 consecutive columns per lane at 64 / 128 outputs: 0 = from 65536 rows,
                               * 1 = never, 2 = always (A/B measurements) */
  int32_t rowgemm_rt;         /* forward row-streaming unary kernels with the weights in registers over two row tiles per wave:
                               * 0 = from 4096 rows at reductions of 16 / 32 / 64, 1 = never (A/B measurements) */
  int32_t reserved[6];
} d3f_tunables;
void d3f_get_tunables(d3f_tunables* out);
int d3f_set_tunables(const d3f_tunables* in);

/* ------------------------------------------------------------------------------------------------
 * Radius neighbors -- replaces radius_neighbors.batch_query
 *   (cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238 -> neighbors/neighbors.cpp:211-333) and the column
 *   truncation of datasets/dataloader.py:52-67 (batch_neighbors_kpconv).
 * One uniform cell list ("grid") is built per support cloud + radius and can serve several query sets
 * (conv, pool and upsample searches of one pyramid level share it).
 * ---------------------------------------------------------------------------------------------- */
size_t d3f_radius_grid_ws_bytes(int Ns);
/* Build the cell list of `supports` for `radius` into grid_ws. */
int d3f_radius_grid_build(const float* supports, int Ns, const int32_t* s_len, int B, float radius,
                          void* grid_ws, size_t grid_ws_bytes, int32_t* status, void* stream);
/* A pyramid build clears the bucket counters of its five cell lists (the first d3f_radius_grid_zero_bytes(Ns) bytes of each
 * grid_ws) and its per-table counters with ONE launch (d3f_zero_buffers, up to 8 buffers of 4-byte multiples) and builds
 * the lists with d3f_radius_grid_build_prezeroed. */
size_t d3f_radius_grid_zero_bytes(int Ns);
int d3f_zero_buffers(void* const* ptrs, const size_t* bytes, int n, void* stream);
/* One launch for the inputs of a (stacked) training step -- the dataset item of reference datasets/ThreeDMatch.py:135-149:
 * kinds[j] = 0 copies bytes[j] (4-byte multiple) srcs[j] -> dsts[j] on the device; 1 writes the circle loss's mask
 * dsts[j][i] (uint8) = srcs[j][i] (float64 dist_keypts) > threshold (utils/loss.py:119), i < bytes[j] / 8.  n <= 24. */
int d3f_copy_buffers(const void* const* srcs, void* const* dsts, const size_t* bytes, const int* kinds, int n,
                     double threshold, void* stream);
/* d3f_radius_query_prefix for the rows nobody has filled yet: a row q with done_rows[q] > 0 is left untouched. */
int d3f_radius_query_prefix_missing(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len, int Ns,
                                    const int32_t* s_len, int B, float grid_radius, float radius, float prefix_radius,
                                    float nearest_bound, int width, int32_t* out_idx, const int32_t* done_rows,
                                    int32_t* status, void* stream);
/* A pooling search (coarse queries over the fine cloud, reference datasets/dataloader.py:141-146) that leaves its TRANSPOSE
 * behind: capped table, max count(s) and last kept keys as d3f_radius_query_ex, and every fine point f found within the
 * radius of coarse query c gets the key (d2 bits << 32 | c) appended to tr_keys[32 f ...] (scratch of 32 Ns uint64),
 * tr_counts[f] [Ns] int32 (cleared by the caller) counting them.  These are the pairs of the upsampling search at the same
 * radius seen from the fine side, with the same distance bits.  A list that outgrows its 32 slots keeps counting (the
 * ranking drops it and the row is searched for). */
int d3f_radius_query_pool_transposed(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len, int Ns,
                                     const int32_t* s_len, int B, float grid_radius, float radius, int width,
                                     int32_t* out_idx, int32_t* max_count, uint64_t* out_last_key, int max_count_group,
                                     int32_t* tr_counts, uint64_t* tr_keys, int32_t* status, void* stream);
/* The training engine's upsampling rows (prefix form: the coarse points within the POOLING radius of every fine point,
 * ranked by (d2, index): dataloader.py:147-152 restricted to what closest_pool, models/blocks.py:79-91, and the transposed
 * pooling table read) ranked from those lists: rows of `up` [Nf, width] (shadow = Nc) with at least one key are written;
 * rows with counts[f] == 0 (the fine point's own voxel barycentre lies farther than the pooling radius; padding rows) are
 * for d3f_radius_query_prefix_missing, and so are the rows whose list outgrew its 32 slots: counts[f] is set back to 0. */
int d3f_upsample_rows_rank(int32_t* counts, const uint64_t* keys, int Nf, int Nc, int width, int32_t* up,
                           void* stream);
int d3f_radius_grid_build_prezeroed(const float* supports, int Ns, const int32_t* s_len, int B, float radius,
                                    void* grid_ws, size_t grid_ws_bytes, int32_t* status, void* stream);
/* Query: out_idx [Nq,width] gets, per query, the in-radius supports of the same batch element, ordered by
 * (d2, index) ascending, first `width` kept, padded with Ns.  out_counts [Nq] (optional) = uncapped count;
 * max_count (optional, 1 int32, caller-zeroed) = max over queries.  d2 arithmetic and the strict d2 < r2 test
 * follow nanoflann.hpp:433-441,249-251 bit for bit. */
int d3f_radius_query(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len,
                     const float* supports, int Ns, const int32_t* s_len, int B, float radius, int width,
                     int32_t* out_idx, int32_t* out_counts, int32_t* max_count, int32_t* status, void* stream);

/* Extended query.  radius <= grid_radius: a search with a smaller radius on a cell list built for grid_radius (the
 * 27-cell scan is a superset).  Optional extra outputs, for the gather-form KPConv grad-input (no reference
 * counterpart -- autograd scatters):
 *   out_wide [Nq, wide_width]: the WHOLE ranked list of every query, padded with Ns (more than wide_width entries
 *     sets D3F_ST_WIDE_OVERFLOW);
 *   out_last_key [Nq] uint64: rank key (d2 bits << 32 | index) of the last entry the capped row (width) keeps, ~0 when
 *     the row keeps every candidate.  s is listed by query q  <=>  d2(q,s) < r2 and key(q,s) <= last_key[q], which
 *     makes the wide list of a point s over the QUERY cloud, filtered by that test, the transpose of the capped
 *     table (the in-radius relation is symmetric and d2 is bit-identical both ways).
 *   max_count_group > 0: max_count is an array of ceil(B / max_count_group) words, one per group of that many
 *     consecutive clouds (8 stacked pairs -> the max count each pair's own table would have had); 0: one word. */
int d3f_radius_query_ex(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len, int Ns,
                        const int32_t* s_len, int B, float grid_radius, float radius, int width, int32_t* out_idx,
                        int32_t* out_counts, int32_t* max_count, int32_t* out_wide, int wide_width,
                        uint64_t* out_last_key, int max_count_group, int32_t* status, void* stream);
/* Prefix form of a search, for the upsampling tables INSIDE the training engine (datasets/dataloader.py:148-150 builds
 * them with radius 2 r; the network reads column 0, models/blocks.py:79-91, and this build reads the transpose of the
 * pooling table off their leading part): row q = the supports within prefix_radius of q, ranked exactly like the leading
 * part of the d3f_radius_query row, or -- when there is none -- the single nearest support within `radius`.  The entries
 * between the two radii are neither ranked nor stored.  collate_fn_descriptor's tables keep the full rows.
 * nearest_bound (0: none; else prefix_radius <= nearest_bound <= radius): the caller's guarantee that every query has a
 * support within that distance -- a fine point lies within the voxel diagonal 0.8 r sqrt(3) < 1.5 r of its own voxel's
 * barycentre, the coarse point it was subsampled into (dataloader.py:141-146) -- so cells beyond it are not scanned. */
int d3f_radius_query_prefix(const void* grid_ws, const float* queries, int Nq, const int32_t* q_len, int Ns,
                            const int32_t* s_len, int B, float grid_radius, float radius, float prefix_radius,
                            float nearest_bound, int width, int32_t* out_idx, int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Grid subsampling -- replaces grid_subsampling.subsample_batch, points-only branch
 *   (cpp_wrappers/cpp_subsampling/wrapper.cpp:62-333 -> grid_subsampling/grid_subsampling.cpp:5-211),
 *   i.e. batch_grid_subsampling_kpconv (datasets/dataloader.py:12-22).
 * order: D3F_ORDER_REFERENCE reproduces the reference's row order (iteration order of the libstdc++
 *   std::unordered_map<size_t,...> it accumulates into, grid_subsampling.cpp:48,85);
 *   D3F_ORDER_FIRST_SEEN emits cells in order of their first input point (cheaper).
 * N is a row CAPACITY: only the first sum(len) rows are read, so levels can be chained without reading
 * lengths back to the host.  Outputs: out_points [out_cap,3] (rows past the emitted total are zero-filled; a total
 * above out_cap sets D3F_ST_CAPACITY), out_len [B], out_total [1].
 * ---------------------------------------------------------------------------------------------- */
#define D3F_ORDER_REFERENCE 0
#define D3F_ORDER_FIRST_SEEN 1
size_t d3f_grid_subsample_ws_bytes(int N, int B);
int d3f_grid_subsample(const float* points, int N, const int32_t* len, int B, float sampleDl, int max_p, int order,
                       float* out_points, int out_cap /* rows of out_points; <= 0: N */, int32_t* out_len,
                       int32_t* out_total, void* ws, size_t ws_bytes, int32_t* status, void* stream);
/* The same call with the reference's optional per-point features [N,fdim] (float) and classes [N,ldim] (int) --
 *   subsample_batch(points, batches, features=, classes=) (wrapper.cpp:75-82,240-247; datasets/dataloader.py:24-50).
 *   out_features [out_cap,fdim] = sequential float32 sum of the cell's member features in input order / (float)count
 *   (grid_subsampling.h:50, .cpp:89-95); out_classes [out_cap,ldim] = per column the value with the most votes, ties
 *   resolved like the reference's std::max_element over its std::unordered_map<int,int> (.cpp:97-102).  Either input
 *   may be NULL (that output is then not written).  Rows follow out_points (same order, same max_p truncation).
 *   NOTE reference bug kept out: with ldim > 1 AND more than one cloud the reference slices the classes of clouds
 *   b > 0 with a wrong end iterator (.cpp:157-158, undefined behaviour); this entry point slices correctly. */
int d3f_grid_subsample_ex(const float* points, int N, const int32_t* len, int B, float sampleDl, int max_p, int order,
                          const float* features, int fdim, const int32_t* classes, int ldim, float* out_points,
                          int out_cap, int32_t* out_len, int32_t* out_total, float* out_features, int32_t* out_classes,
                          void* ws, size_t ws_bytes, int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------------
 * KPConv -- replaces models/blocks.py:237-382 (KPConv.forward, rigid / 'linear' / 'sum' path) and its
 *   autograd backward.
 *   out[n,:] = ( sum_k ( sum_h w[n,h,k] * x[idx[n,h],:] ) @ W[k] ) / nn[n]
 *   w = max(0, 1 - |(s[idx[n,h]] - q[n]) - kp[k]| / extent),  nn[n] = max(1, #{h : sum_c x[idx[n,h],c] > 0})
 * nn_out [Nq] float32 is saved for the backward pass.
 * wf_save (optional, [Nq, d3f_kpconv_saves_wf(...)] float32): the weighted features sum_h w[n,h,k] x[idx[n,h],c] are
 *   left there for the backward pass (what autograd keeps alive in the reference as `weighted_features`,
 *   blocks.py:375).  d3f_kpconv_saves_wf returns the floats per query: K*Cin, except for the Cin <= 4 input-layer
 *   kernels, whose rows are padded to 16 kernel-point slots (16*Cin; slots >= K are zero), and 0 when nothing is saved.
 * spack_keep (optional, 16*Ns bytes) / grad_x_clear (optional, [Ns, Cin]): the forward packs the supports as
 *   float4 {x, y, z, [sum_c feat > 0]}; with spack_keep the packed array is left in the caller's buffer and handed
 *   back to the backward pass (spack_kept), which then launches no packing kernel of its own; grad_x_clear is the
 *   backward's scatter target, cleared here on the side (pass grad_x_precleared = 1 to the backward).  Honoured when
 *   d3f_kpconv_packs_supports says so (otherwise pass NULL / 0).
 *   grad_x_clear == D3F_SPACK_READY: spack_keep ALREADY holds the packed supports of (s_pts, x) -- written by the
 *   epilogue that produced x (d3f_bias_act_forward, spack_out) -- and whatever needed clearing was cleared there; the
 *   forward then launches no packing kernel at all.  Same convention for d3f_kpconv_aggregate.
 * ---------------------------------------------------------------------------------------------- */
#define D3F_SPACK_READY ((float*)(uintptr_t)1)
int d3f_kpconv_forward(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                       const float* x, int Cin, const float* kernel_points, int K, const float* weights, int Cout,
                       float extent, float* out, float* nn_out, float* wf_save, void* spack_keep, float* grad_x_clear,
                       void* ws, size_t ws_bytes, void* stream);
int d3f_kpconv_saves_wf(int Cin, int Cout, int K, int H);
int d3f_kpconv_packs_supports(int Cin, int Cout, int K, int H, int Ns);
size_t d3f_kpconv_ws_bytes(int Nq, int Ns, int H, int K, int Cin, int Cout);
/* grad_x [Ns,Cin] and grad_w [K,Cin,Cout] are OVERWRITTEN.  wf_saved (optional): the forward's wf_save; without it
 * the aggregation is recomputed. */
int d3f_kpconv_backward(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                        const float* x, int Cin, const float* kernel_points, int K, const float* weights, int Cout,
                        float extent, const float* nn, const float* grad_out, const float* wf_saved,
                        const void* spack_kept, int grad_x_precleared, float* grad_x, float* grad_w, void* ws,
                        size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Reverse neighbor table + gather-form grad-input of KPConv -- replaces the scatter-add (index_add_) autograd runs
 *   for the gathers of models/blocks.py:277-280,356-359 in backward.
 * d3f_reverse_table_build: CSR transpose of idx [Nq,H] over Ns supports (entries outside [0,Ns) are shadow):
 *   rev_ent[rev_ptr[s] .. rev_ptr[s+1]) = the queries that list support s, ascending; rev_ptr [Ns+1], rev_ent [Nq*H].
 *   Deterministic.  Built once per table (next to the radius search) and reused by every layer on that table.
 * d3f_kpconv_grad_input_gather: grad_x [Ns,Cin] = sum_k (sum_{q in rev(s)} w(q,s,k) grad_out[q,:]/nn[q]) @ W[k]^T
 *   (OVERWRITTEN; no atomics, bit-reproducible).  nn may be NULL (grad_out already divided).  rev(s) comes in one of
 *   three forms: the exact form rev_rel (below), CSR (rev_ptr + rev_ent, rev_last_key NULL) or the search's own output
 *   (rev_ptr NULL): rev_ent =
 *   out_wide [Ns, rev_width] of a d3f_radius_query_ex of the SUPPORT points over the QUERY cloud with the table's
 *   radius, rev_last_key = out_last_key [Nq] of the query that produced the table -- no transposition pass at all.
 *   rev_radius > 0: rev_ent comes from a search with a LARGER radius (a pooling table's transpose is the prefix, within
 *   the pooling radius rev_radius, of the rows of the upsampling table the pyramid holds anyway: radius 2r, ranked by
 *   distance); a full row whose last entry is still within rev_radius sets D3F_ST_WIDE_OVERFLOW in status (optional).
 * ---------------------------------------------------------------------------------------------- */
size_t d3f_reverse_table_ws_bytes(int Nq, int H, int Ns);
int d3f_reverse_table_build(const int32_t* idx, int Nq, int H, int Ns, int32_t* rev_ptr, int32_t* rev_ent, void* ws,
                            size_t ws_bytes, void* stream);
int d3f_kpconv_grad_input_gather_supported(int Cin, int Cout, int K);
int d3f_kpconv_grad_input_gather(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* rev_ptr,
                                 const int32_t* rev_ent, const uint64_t* rev_last_key, int rev_width, float rev_radius,
                                 const float* rev_rel, const float* kernel_points, int K, const float* weights, int Cin,
                                 int Cout, float extent, const float* nn, const float* grad_out, float* grad_x,
                                 int32_t* status, void* stream);
/* d3f_reverse_table_filter: search form -> EXACT form, once per table in the pyramid build: rev_rel_out [Ns, rev_width]
 * float4 {q - s, bits of q} = the entries of row s that pass the membership test (key <= rev_last_key[q], and d2 <
 * rev_radius^2 when rev_radius > 0), compacted in rank order, rows padded with index Nq.  d3f_kpconv_grad_input_gather
 * with rev_rel (rev_ptr / rev_ent / rev_last_key NULL) then reads each neighborhood as one coalesced run: no position
 * or key gathers, no membership test on the training stream. */
int d3f_reverse_table_filter(const int32_t* rev_ent, int rev_width, const uint64_t* rev_last_key, const float* q_pts,
                             int Nq, const float* s_pts, int Ns, float rev_radius, float* rev_rel_out, int32_t* status,
                             void* stream);

/* grad_x alone, from gwf = (grad_out / nn) @ W^T  [Nq, K*Cin] computed by the caller (an ordinary GEMM: the right
 * tool for the few-point / 256..512-channel layers at the bottom of the U-Net, where the fused kernel's own gW tile
 * would run on a handful of workgroups).  Same semantics as the grad_x of d3f_kpconv_backward.
 * Workspace: d3f_kpconv_ws_bytes(Nq, Ns, H, K, Cin, 64). */
int d3f_kpconv_grad_input_supported(int Cin, int K, int H, int Ns);
/* forward counterpart for the same layers: only the neighbor aggregation wf [Nq, K*Cin] and nn [Nq]; the caller
 * computes out = (wf @ W) / nn with a GEMM (and d3f_bias_act_forward's row_div). */
int d3f_kpconv_aggregate(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                         const float* x, int Cin, const float* kernel_points, int K, float extent, float* wf_out,
                         float* nn_out, void* spack_keep, float* grad_x_clear, void* ws, size_t ws_bytes,
                         void* stream);
int d3f_kpconv_grad_input(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                          const float* x, int Cin, const float* kernel_points, int K, float extent, const float* gwf,
                          const void* spack_kept, int grad_x_precleared, float* grad_x, void* ws, size_t ws_bytes,
                          void* stream);
/* The aggregation of d3f_kpconv_aggregate on the TRANSPOSED graph (autograd of models/blocks.py:359-380 with the sums
 * over queries and over (kernel point, output channel) exchanged):
 *   agg_out [Ns, K*Cout],  agg_out[s, k, o] = sum_{q lists s} w(q, s, k) * grad_out[q, o] (/ nn[q] when nn != NULL)
 * over the exact-form reverse table of d3f_reverse_table_filter (rev_rel [Ns, rev_width, 4]); the caller's GEMM
 *   grad_x [Ns, Cin] = agg_out @ W',  W'[k*Cout + o, c] = weights[k, c, o]
 * finishes the grad-input without atomics (every row written once, bit-reproducible).  Registers -> HBM, no LDS tile:
 * the wide layers (>= 64 channels), whose contraction dominates, run it as a tall library GEMM instead of inside the
 * fused gather kernel.  Every row of agg_out is written (rows without reverse neighbors get zeros). */
int d3f_kpconv_aggregate_transposed_supported(int Cout, int K);
int d3f_kpconv_aggregate_transposed(const float* rev_rel, int rev_width, int Ns, int Nq, const float* kernel_points,
                                    int K, float extent, const float* nn, const float* grad_out, int Cout,
                                    float* agg_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Weight gradient of the 1x1 "unary" convolutions -- replaces autograd's grad_out^T @ x for nn.Linear in
 * UnaryBlock (models/blocks.py:481-515): grad_w [Cout, Cin] = grad_out^T [Cout, N] @ x [N, Cin], with the
 * reduction over the N points spread over the chip (deterministic two-pass sum, no atomics).
 * Supported when Cin and Cout are multiples of 16 (d3f_linear_grad_weight_supported).
 * ---------------------------------------------------------------------------------------------- */
int d3f_linear_grad_weight_supported(int N, int Cin, int Cout);
/* The same blocks at the upper pyramid levels (many rows, Cin/Cout <= 256): x W^T with the epilogue
 * act(. + bias1 + add + bias2) applied before the only store, and grad_x = grad_out W, as row-streaming f32-MFMA
 * kernels (a library GEMM is launch-bound on these shapes).  Supported: Cin, Cout in {32, 64, 128, 256} on the output
 * side, a multiple of 16 up to 1024 on the reduction side (d3f_linear_fused_supported checks both directions). */
int d3f_linear_fused_supported(int N, int Cin, int Cout);
int d3f_linear_bias_act_forward(const float* x, const float* weight, int N, int Cin, int Cout, const float* bias1,
                                const float* add, const float* bias2, float slope, float* out, float* zero_init,
                                int zero_n, void* stream);
/* The last unary block of a bottleneck AND its shortcut unary in one launch (models/blocks.py:658-686: unary2(x),
 * unary_shortcut(features), leaky_relu(x + shortcut)): out = act(x1 w1^T + x2 w2^T + bias1 + bias2 + bias3 + bias4); the
 * [N, Cout] shortcut tensor is never formed.  Served while both weight matrices stay in registers
 * (d3f_linear_pair_supported: from 4096 rows, (Cin1 | Cin2) -> Cout = (32 | 64) -> 128 or (16 | 32) -> 64). */
int d3f_linear_pair_supported(int N, int Cin1, int Cin2, int Cout);
int d3f_linear_pair_bias_act_forward(const float* x1, const float* w1, int Cin1, const float* x2, const float* w2, int Cin2,
                                     int N, int Cout, const float* bias1, const float* bias2, const float* bias3,
                                     const float* bias4, float slope, float* out, float* zero_init, int zero_n,
                                     void* stream);
/* grad_x [N,Cin] = grad_out [N,Cout] @ weight [Cout,Cin] (+ add [N,Cin] when given: the gradient another branch
 * produced for the same tensor, accumulated in the epilogue instead of by a separate launch) */
int d3f_linear_grad_input(const float* grad_out, const float* weight, int N, int Cin, int Cout, const float* add,
                          float* grad_x, void* stream);
/* W'[k][o][c] = W[k][c][o] for up to 16 KPConv weight tensors [K, Cin, Cout] in ONE launch (Cin, Cout multiples of 32):
 * the permuted matrices the transposed-aggregation grad-input contracts with -- autograd's grad of
 * torch.matmul(weighted_features, self.weights) (models/blocks.py:369-374) w.r.t. the features, seen from the supports.
 * All pointer / size arrays are HOST arrays of n entries; the tensors are device memory. */
int d3f_permute_kpconv_weights(const float* const* srcs_host, float* const* dsts_host, const int* K_host,
                               const int* Cin_host, const int* Cout_host, int n, void* stream);
/* The contractions of the wide / few-row layers with their epilogue, y [R,N] = act(x [R,K] . B / row_div + bias1 + add +
 * bias2), as ONE f32-MFMA launch (two when the reduction is split: few rows against a long reduction) -- what the
 * reference computes as torch.matmul followed by separate bias / residual / LeakyReLU ops:
 *   KPConv's `torch.matmul(weighted_features, self.weights)` summed over kernel points (models/blocks.py:362-374) with
 *   the /neighbor-count of :376-380 and the bias + LeakyReLU of :473,:598,:676 (mode 1, B = weights viewed [K Cin, Cout],
 *   row_div = neighbor counts); nn.Linear of UnaryBlock (:481-541, y = x W^T: mode 0, B = weight [N, K]) with the
 *   residual add of :686; and autograd's grad-input products g W (mode 1) / g W^T (mode 0).
 * mode 0: B [N, ldw] reduction-contiguous (rows = output columns); kblock > 0: B is [K / kblock][N][kblock] -- the
 *   reduction index b kblock + o of output column n lives at w + (b N + n) kblock + o, i.e. KPConv's weights
 *   [K, Cin, Cout] read as the permuted matrix W'[k, o, c] = W[k, c, o] of the transposed-aggregation grad-input
 *   (ldw is ignored).
 * mode 1: B [K, ldw] (rows = reduction indices).
 * K and N multiples of 16; x, w, y, add, biases 16-byte aligned, leading dimensions multiples of 4.  row_div / bias1 /
 * add [R, ldadd] / bias2 optional, slope = 1: no activation.  zero_init / zero_n as in d3f_bias_act_forward.
 * ws >= d3f_gemm_epilogue_ws_bytes(R, K, N) (slabs of a split reduction; may be 256 bytes).  Bit-reproducible. */
int d3f_gemm_epilogue_supported(int R, int K, int N, int mode, int kblock);
size_t d3f_gemm_epilogue_ws_bytes(int R, int K, int N);
int d3f_gemm_epilogue(const float* x, int ldx, const float* w, int ldw, int mode, int kblock, int R, int K, int N,
                      const float* row_div, const float* bias1, const float* add, int ldadd, const float* bias2,
                      float slope, float* y, int ldy, float* zero_init, int zero_n, void* ws, size_t ws_bytes,
                      void* stream);
size_t d3f_linear_grad_weight_ws_bytes(int N, int Cin, int Cout);
int d3f_linear_grad_weight(const float* x, const float* grad_out, int N, int Cin, int Cout, float* grad_w, void* ws,
                           size_t ws_bytes, void* stream);
/* The same, and the second-stage launch (the fixed-order sum of the partial gradient slabs) also finishes the bias
 * gradient of the block -- autograd's grad_out.sum(0) for the bias of nn.Linear / BatchNormBlock's bias
 * (models/blocks.py:473,497): grad_bias[c] (and grad_bias2[c], optional) = sum_b bias_part[b][c] over the
 * bias_blocks x bias_cols partial column sums d3f_bias_act_backward_partial wrote. */
int d3f_linear_grad_weight_bias(const float* x, const float* grad_out, int N, int Cin, int Cout, float* grad_w, void* ws,
                                size_t ws_bytes, const float* bias_part, int bias_blocks, int bias_cols,
                                float* grad_bias, float* grad_bias2, void* stream);
/* ALL weight gradients of a backward stage in two launches (round 6).  A weight gradient has no consumer before the
 * optimizer (reference trainer.py:103-111: loss.backward() completes, then optimizer.step()), so the autograd nodes of
 * the host mirror only QUEUE their problem -- the nn.Linear weights of the unary blocks (models/blocks.py:481-515,
 * autograd's grad_out^T @ x) and the KPConv weights (blocks.py:369-374, with x := g / nn [Nq, Cout],
 * grad_out := weighted features [Nq, K Cin], grad_w viewed as [K Cin, Cout]) -- and the stage ends with ONE call:
 * one launch walks every problem's (row partition, output block) tasks, one more sums every problem's slabs in a fixed
 * order and finishes the queued bias gradients (as d3f_linear_grad_weight_bias).  Bit-reproducible, no atomics.
 * problems / n: HOST array.  grad_w [Cout, ldw]: row stride ldw >= Cin (a column block of a wider weight matrix is
 * written in place).  x, grad_out 16-byte aligned; Cin, Cout multiples of 16.  bias_part == NULL: no bias gradient. */
typedef struct d3f_atb_problem {
  const float* x;        /* [N, Cin] */
  const float* grad_out; /* [N, Cout] */
  float* grad_w;         /* [Cout, ldw] */
  int32_t N, Cin, Cout, ldw;
  const float* bias_part;
  int32_t bias_blocks, bias_cols;
  float* grad_bias;
  float* grad_bias2;
} d3f_atb_problem;
size_t d3f_linear_grad_weight_group_ws_bytes(const d3f_atb_problem* problems_host, int n);
int d3f_linear_grad_weight_group(const d3f_atb_problem* problems_host, int n, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pools -- replace models/blocks.py:94-110 (max_pool) and :79-91 (closest_pool).
 * ---------------------------------------------------------------------------------------------- */
/* out[n,c] = max_h x'[idx[n,h],c], x' = x plus a zero shadow row; argmax_out [Nq,C] int32 = winning support row
 * (Ns if the shadow won), saved for backward.
 * grad_x_clear (optional, [Ns,C]): the backward's scatter target, zeroed by the forward launch on the side; the backward
 * is then called with grad_x_precleared = 1 and launches no fill. */
/* width_dev (optional, device int32[1]): the table's max neighbor count as d3f_radius_query reports it; only the
 * first min(H, *width_dev) columns take part, which is the table the reference would have built
 * (dataloader.py:64-66 trims to the max count) when idx is kept at a wider, static width.
 * q_len (optional, [B]) + group: the batch stacks several reference batches (groups of `group` consecutive clouds, e.g.
 * 8 pairs); width_dev is then an array with one entry per group and every query row uses its own group's. */
int d3f_max_pool_forward(const float* x, int Ns, int C, const int32_t* idx, int Nq, int H, float* out,
                         int32_t* argmax_out, float* grad_x_clear, const int32_t* width_dev, const int32_t* q_len,
                         int B, int group, void* stream);
int d3f_max_pool_backward(const float* grad_out, const int32_t* argmax, int Nq, int C, int Ns, float* grad_x,
                          int grad_x_precleared, void* stream);
/* out[n,:] = x'[idx[n,0],:]  (idx has row stride H); backward: grad_out has row stride ld >= C (a column slice of
 * the gradient of the decoder's concatenation is consumed in place) */
/* skip (optional, [Nq, Cs]): out is [Nq, C + Cs] = [upsampled | skip], the decoder's torch.cat([x, skip], dim=1)
 * (models/architectures.py:311-313) done by the same launch; Cs = 0: plain closest_pool. */
int d3f_closest_pool_forward(const float* x, int Ns, int C, const int32_t* idx, int Nq, int H, const float* skip,
                             int Cs, float* out, float* grad_x_clear, void* stream);
int d3f_closest_pool_backward(const float* grad_out, int ld, const int32_t* idx, int Nq, int H, int C, int Ns,
                              float* grad_x, int grad_x_precleared, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Block epilogue -- replaces BatchNormBlock's bias add (models/blocks.py:473, use_bn=False), nn.LeakyReLU(0.1)
 *   (:497,:598,:676), the bottleneck's residual add (:686) and, in backward, PyTorch's bias-gradient reduction.
 *   out = act(x + bias1 + (add + bias2)), act(v) = v > 0 ? v : slope*v  (slope = 1: identity); bias1/add/bias2 optional.
 *   backward: grad_x = grad_out * (out > 0 ? 1 : slope)  (also the gradient of `add`);
 *             grad_bias[c] = sum_n grad_x[n,c]  (gradient of bias1 and bias2; OVERWRITTEN); grad_bias2 (optional)
 *             receives the same sums in a second buffer (two parameters, identical gradients).
 * zero_init (optional, zero_n floats): scratch the forward kernel clears on the side -- the backward's bias-gradient
 *   accumulators, so that backward (bias_prezeroed = 1) needs no separate fill launch.
 * row_div (optional, [N]): out = act(x[n,:]/row_div[n] + ...) and grad_x[n,:] = masked gradient / row_div[n] (the bias
 *   sums use the undivided masked gradient) -- the KPConv neighbor-count normalisation when x is a raw wf @ W product.
 * add_idx (optional, int32 with row stride idx_stride): `add` is then a COARSE matrix [add_rows, C] and row n receives
 *   add[add_idx[n*idx_stride]] (zeros for a shadow index) -- nearest upsampling folded into the epilogue.
 * ---------------------------------------------------------------------------------------------- */
int d3f_bias_act_forward(const float* x, const float* bias1, const float* add, const float* bias2, float slope, int N,
                         int C, float* out, float* zero_init, int zero_n, const float* row_div, const int32_t* add_idx,
                         int idx_stride, int add_rows, void* stream);
/* The same launch, additionally preparing the KPConv that consumes `out` as its input features: spack_out [N] float4 =
 * {s_pts[n], (sum_c out[n,c] > 0)} (the packed supports d3f_kpconv_forward would otherwise build with a launch of its
 * own) and zero_like_out (optional, [N,C]) cleared (that KPConv's grad_x scatter target).  C in {16,...,512} with C/4 a
 * power of two (d3f_bias_act_packs(C)). */
int d3f_bias_act_packs(int C);
int d3f_bias_act_forward_pack(const float* x, const float* bias1, const float* add, const float* bias2, float slope,
                              int N, int C, float* out, float* zero_init, int zero_n, const float* row_div,
                              const int32_t* add_idx, int idx_stride, int add_rows, const float* s_pts,
                              void* spack_out, float* zero_like_out, void* stream);
/* ws (optional, d3f_bias_act_backward_ws_bytes): with it, N >= 4096 uses per-block partial sums + a second tiny
 * launch for the bias gradient instead of atomics on C addresses (which serialise: 32 us at 38k x 32), deterministic. */
size_t d3f_bias_act_backward_ws_bytes(int N, int C);
int d3f_bias_act_backward(const float* grad_out, const float* out, float slope, int N, int C, float* grad_x,
                          float* grad_bias, float* grad_bias2, int bias_prezeroed, const float* row_div, void* ws,
                          size_t ws_bytes, void* stream);
/* The two passes of the many-row form separately (N >= 4096: d3f_bias_act_backward_blocks(N, C) > 0 partial rows):
 * _partial writes grad_x (optional) and the partial column sums ws [blocks, C]; the bias gradient is finished either by
 * d3f_linear_grad_weight_bias -- inside the launch that sums the weight gradient's slabs -- or by d3f_bias_sum. */
int d3f_bias_act_backward_blocks(int N, int C);
int d3f_bias_act_backward_partial(const float* grad_out, const float* out, float slope, int N, int C, float* grad_x,
                                  const float* row_div, void* ws, size_t ws_bytes, void* stream);
int d3f_bias_sum(const float* part, int blocks, int C, float* grad_bias, float* grad_bias2, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Deformable KPConv -- replaces the deformable=True branch of KPConv.forward (models/blocks.py:243-257,286-324,365-366).
 *   kp_def [Nq,K,3] = offsets * 1 + kernel_points, the per-query kernel points (offsets come from the rigid
 *   `offset_conv`, :190-199,244-254, which runs on d3f_kpconv_forward).  A neighbor is live when it is a real support
 *   AND within `extent` of at least one deformed kernel point (extent_sq = the float32 the reference compares against,
 *   (float)(KP_extent**2), :304); dead neighbors count neither for the weights nor for the neighbor number (:304-321).
 *   aggregate: wf [Nq,K*Cin] = sum_{h live} w_mode x[idx],  nn [Nq] = max(1, #live neighbors with a positive feature
 *              sum), min_d2 [Nq,K] = min_h d2[n,h,k] and min_idx [Nq,K] = the support attaining it (Ns when the row has
 *              no real neighbor; min_d2 is then the distance to the reference's shadow point at 1e6) -- both optional.
 *   grad:      from gwf = dL/dwf: grad_x [Ns,Cin] (OVERWRITTEN; optional) and grad_kp [Nq,K,3] (optional), the
 *              gradient w.r.t. the deformed kernel points through the influence weights.
 *   The caller applies the modulations, the contraction with the kernel weights and the division by nn.
 *   mode as d3f_kpconv_aggregate_modes.
 * ---------------------------------------------------------------------------------------------- */
int d3f_kpconv_deform_aggregate(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                                const float* x, int Cin, const float* kp_def, int K, float extent, float extent_sq,
                                int mode, float* wf_out, float* nn_out, float* min_d2_out, int32_t* min_idx_out,
                                void* stream);
int d3f_kpconv_deform_grad(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                           const float* x, int Cin, const float* kp_def, int K, float extent, float extent_sq, int mode,
                           const float* gwf, float* grad_x, float* grad_kp, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Batch normalisation over the stacked points -- replaces the use_bn=True branch of BatchNormBlock
 *   (models/blocks.py:454-455,465-471: nn.BatchNorm1d(C, momentum) over x [N, C] viewed as [1, C, N]).
 *   training != 0: y = (x - mean) / sqrt(var + eps) * gamma + beta with the batch mean and BIASED batch variance;
 *                  running_mean / running_var (optional) are updated in place: (1-m) old + m new, the variance
 *                  UNBIASED (N/(N-1)), like torch.
 *   training == 0: the running statistics normalise (both required).
 *   slope: LeakyReLU fused behind (1 = none).  gamma / beta may be NULL (1 / 0).  save_mean / save_invstd [C]
 *   (optional forward outputs) are what the backward needs.  n_live (optional int32 device scalar): only the first
 *   min(N, *n_live) rows are live -- N is then a capacity; dead rows are written as zeros.
 *   backward: grad_x (optional), grad_gamma, grad_beta (optional, OVERWRITTEN) for grad_y taken behind the activation.
 * Deterministic (fixed-order partial sums in ws).
 * ---------------------------------------------------------------------------------------------- */
size_t d3f_batchnorm_ws_bytes(int N, int C);
int d3f_batchnorm_forward(const float* x, int N, int C, const int32_t* n_live, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, int training,
                          float slope, float* y, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                          void* stream);
int d3f_batchnorm_backward(const float* x, int N, int C, const int32_t* n_live, const float* gamma, const float* beta,
                           const float* save_mean, const float* save_invstd, float slope, int training,
                           const float* grad_y, float* grad_x, float* grad_gamma, float* grad_beta, void* ws,
                           size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Detector score -- replaces KPFCNN.detection_scores (models/architectures.py:322-368).
 * feat [N,C] un-normalised descriptors, idx = neighbors[0] [N,H].  feat_max [1] is a device scalar holding
 * max(feat) (d3f_global_max computes it).  training != 0: soft score; training == 0 additionally applies the
 * local-maximum gate (:361-366).
 * ---------------------------------------------------------------------------------------------- */
int d3f_global_max(const float* x, size_t n, float* out_max, void* ws /* >= 4 bytes */, size_t ws_bytes, void* stream);
/* same over the first sum(len) rows of x [cap_rows, C] (row count read on the device) */
int d3f_global_max_rows(const float* x, int cap_rows, int C, const int32_t* len, int B, float* out_max, void* ws,
                        size_t ws_bytes, void* stream);
/* one maximum per group of `group` consecutive clouds: out_max [ceil(B/group)] (ws >= 4 bytes per group).  The
 * reference normalises the detector features by the maximum of ITS batch (one pair, architectures.py:342); with 8
 * pairs stacked the normaliser has to stay per pair. */
int d3f_global_max_groups(const float* x, int cap_rows, int C, const int32_t* len, int B, int group, float* out_max,
                          void* ws, size_t ws_bytes, void* stream);
/* aux (optional, training only, [N, d3f_detection_scores_aux_floats(C)]): per-point scalars of the winning channel,
 * left behind so that the backward pass does not gather features again. */
int d3f_detection_scores_aux_floats(int C); /* 8 for C in {16, 32, 64}, 0 (aux unsupported) otherwise */
/* width (optional, device int32[1]): max neighbor count of the table when idx is kept wider than the reference would
 * build it (min(limit, max_count) columns, dataloader.py:64-66): the extra all-shadow columns are then ignored -- they
 * would give the eval-mode local-maximum gate a zero candidate the reference does not have.
 * len (optional, [B]) + group: stacked reference batches (see d3f_max_pool_forward): feat_max and width are arrays with
 * one entry per group (d3f_global_max_groups; d3f_radius_query_ex max_count_group). */
int d3f_detection_scores_forward(const float* feat, int N, int C, const int32_t* idx, int H, const float* feat_max,
                                 int training, float* scores, float* aux, const int32_t* width, const int32_t* len,
                                 int B, int group, void* stream);
/* grad_feat [N,C] is OVERWRITTEN.  Includes the gradient through the global max normaliser.  aux: the forward's
 * (optional; without it the neighborhood statistics are recomputed). */
int d3f_detection_scores_backward(const float* feat, int N, int C, const int32_t* idx, int H, const float* feat_max,
                                  const float* grad_scores, const float* aux, float* grad_feat, void* ws,
                                  size_t ws_bytes, void* stream);
/* the same for stacked reference batches (len [B] + group as in the forward; feat_max [ceil(B/group)]): the gradient
 * through the normaliser stays inside each group -- P fragment pairs stacked into one TRAINING batch keep the
 * per-pair maximum of architectures.py:342 and its arg-max gradient.  ws >= d3f_detection_scores_ws_bytes (per-block partial
 * sums of every group, added up in a fixed order: no float atomics in the normaliser's gradient).  A point whose incoming
 * gradient is exactly 0 -- all but the correspondences of the detector loss, utils/loss.py:140-158 -- contributes nothing and
 * is skipped. */
int d3f_detection_scores_backward_groups(const float* feat, int N, int C, const int32_t* idx, int H,
                                         const float* feat_max, const float* grad_scores, const float* aux,
                                         float* grad_feat, const int32_t* len, int B, int group, void* ws,
                                         size_t ws_bytes, void* stream);
size_t d3f_detection_scores_ws_bytes(int N, int C);

/* ------------------------------------------------------------------------------------------------
 * Descriptor loss -- replaces utils/loss.py: cdist(:8-44, 'euclidean'), CircleLoss.forward(:111-141),
 * DetLoss.forward(:149-158), fused into one single-workgroup launch (M <= 1024 sampled correspondences).
 * anchor/positive [M,C]; neg_mask [M,M] uint8 = (dist_keypts > safe_radius), evaluated by the caller in the
 * dtype the dataset supplies (float64 in the reference, loss.py:116); anc_score/pos_score [M].
 * Outputs: dists [M,M], furthest_positive [M], average_negative [M],
 *   out_scalars[0..5] = desc_loss, det_loss, accuracy(%), mean furthest_positive, mean average_negative,
 *                      desc_loss + det_loss;
 *   stats [d3f_circle_det_loss_stats_floats(M)] = row/column log-sum-exps + closest negatives, kept for backward.
 * backward: gradients of  grad_desc*desc_loss + grad_det*det_loss  (device scalars; either may be NULL = 0)
 *   wrt anchor, positive [M,C] and the two score vectors [M] (optional).
 * ---------------------------------------------------------------------------------------------- */
size_t d3f_circle_det_loss_stats_floats(int M);
size_t d3f_circle_det_loss_ws_bytes(int M);
int d3f_circle_det_loss_forward(const float* anchor, const float* positive, int M, int C, const uint8_t* neg_mask,
                                const float* anc_score, const float* pos_score, float log_scale, float safe_radius,
                                float pos_margin, float neg_margin, float* dists, float* furthest_positive,
                                float* average_negative, float* out_scalars, float* stats, void* stream);
int d3f_circle_det_loss_backward(const float* anchor, const float* positive, int M, int C, const uint8_t* neg_mask,
                                 const float* anc_score, const float* pos_score, float log_scale, float safe_radius,
                                 float pos_margin, float neg_margin, const float* dists, const float* stats,
                                 const float* grad_desc, const float* grad_det, float* grad_anchor,
                                 float* grad_positive, float* grad_anc_score, float* grad_pos_score, void* ws,
                                 size_t ws_bytes, void* stream);

/* `pairs` fragment pairs stacked into one training batch (the reference trains on one pair per step,
 * datasets/dataloader.py:73; stacking P of them into one launch sequence is this build's batching): every array above
 * gains a leading pair dimension (anchor/positive [pairs*M,C], neg_mask [pairs,M,M], scores [pairs*M], dists
 * [pairs,M,M], out_scalars [pairs,6], stats [pairs * d3f_circle_det_loss_stats_floats(M)]); every pair is its own
 * M x M problem.  out_total [1] = sum_p (w_desc desc_p + w_det det_p): its gradient is the SUM of the pairs' gradients
 * (the caller's optimizer scale makes the mean).  M <= 128, C <= 64, pairs <= 32.  backward: grad_total device scalar. */
int d3f_circle_det_loss_forward_pairs(const float* anchor, const float* positive, int M, int C, int pairs,
                                      const uint8_t* neg_mask, const float* anc_score, const float* pos_score,
                                      float log_scale, float safe_radius, float pos_margin, float neg_margin,
                                      float w_desc, float w_det, float* dists, float* furthest_positive,
                                      float* average_negative, float* out_scalars, float* out_total, float* stats,
                                      void* stream);
int d3f_circle_det_loss_backward_pairs(const float* anchor, const float* positive, int M, int C, int pairs,
                                       const uint8_t* neg_mask, const float* anc_score, const float* pos_score,
                                       float log_scale, float safe_radius, float pos_margin, float neg_margin,
                                       float w_desc, float w_det, const float* dists, const float* stats,
                                       const float* grad_total, float* grad_anchor, float* grad_positive,
                                       float* grad_anc_score, float* grad_pos_score, void* stream);

/* Sampled-correspondence front end of the loss -- replaces F.normalize over all N descriptors
 * (models/architectures.py:318) + the four index selections of trainer.py:91-94 and their backward:
 *   out[m,:] = x[idx[m],:] / max(||x[idx[m],:]||, 1e-12),  s[m] = scores[idx[m]].
 * idx_a / idx_p int64, element m at idx[m * idx_stride] (2 = the columns of the [M,2] correspondence table, read in
 * place); idx_p is offset by *p_offset (device int32: rows of the first cloud) when given.
 * backward: grad_x [N,C] and grad_scores [N] are ONE allocation of N*(C+1) floats (grad_scores == grad_x + N*C),
 * overwritten; g_* may be NULL. */
int d3f_select_normalize_forward(const float* x, const float* scores, int N, int C, const int64_t* idx_a,
                                 const int64_t* idx_p, int idx_stride, int M, const int32_t* p_offset, float* out_a,
                                 float* out_p, float* sa, float* sp, void* stream);
int d3f_select_normalize_backward(const float* x, int N, int C, const int64_t* idx_a, const int64_t* idx_p,
                                  int idx_stride, int M, const int32_t* p_offset, const float* g_a, const float* g_p,
                                  const float* g_sa, const float* g_sp, float* grad_x, float* grad_scores,
                                  void* stream);
/* stacked pairs: clouds 2p, 2p+1 of the stack are pair p (len [2 pairs]: level-0 stack lengths on the device); corr
 * [pairs*M,2] int64 holds every pair's own table with cloud-local rows (trainer.py:91-94 adds len(first cloud) to the
 * second column; here each column is offset by the start of its cloud in the stack); outputs [pairs*M, ...]. */
int d3f_select_normalize_forward_pairs(const float* x, const float* scores, int N, int C, const int64_t* corr, int M,
                                       int pairs, const int32_t* len, float* out_a, float* out_p, float* sa, float* sp,
                                       void* stream);
int d3f_select_normalize_backward_pairs(const float* x, int N, int C, const int64_t* corr, int M, int pairs,
                                        const int32_t* len, const float* g_a, const float* g_p, const float* g_sa,
                                        const float* g_sp, float* grad_x, float* grad_scores, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense matching -- replaces build_correspondence (geometric_registration/common.py:5-21): the
 * [Ns,Nt] distance matrix sqrt(2 - 2 S.T^T) is never materialised; ONE sweep of an f32 MFMA tile kernel over the
 * S x T tiles (the reference forms S T^T once, common.py:9) keeps a running row arg-min in registers and the column
 * arg-min of each workgroup's rows in LDS; the target range is split over workgroups and merged with one 64-bit
 * atomicMin per row / per column on (distance bits, index).  row_argmin [Ns], col_argmin [Nt] int32
 * (lowest index on ties, like np.argmin), mutual [Ns] int32 (optional) = 1 where col_argmin[row_argmin[i]] == i.
 * C in {16, 32, 64, 128}.
 * ---------------------------------------------------------------------------------------------- */
size_t d3f_mutual_nn_ws_bytes(int Ns, int Nt);
int d3f_mutual_nn(const float* src_desc, int Ns, const float* tgt_desc, int Nt, int C, int32_t* row_argmin,
                  int32_t* col_argmin, int32_t* mutual, void* ws, size_t ws_bytes, void* stream);
/* P independent matchings by ONE sweep launch (BASELINE configs[3]: 8 fragment pairs per inference batch).
 * seg [P,4] int32 ON THE DEVICE = {src_off, src_len, tgt_off, tgt_len} per pair: rows of src_desc / tgt_desc (which
 * may be the same stacked matrix).  max_src / max_tgt: host upper bounds of the lengths (they size the grid).
 * row_argmin [src_rows] / col_argmin [tgt_rows] / mutual [src_rows] are indexed by the row of the stacked matrix and
 * hold PAIR-LOCAL indices (what P separate d3f_mutual_nn calls return); rows outside every segment are untouched. */
size_t d3f_mutual_nn_batched_ws_bytes(int src_rows, int tgt_rows);
int d3f_mutual_nn_batched(const float* src_desc, int src_rows, const float* tgt_desc, int tgt_rows, const int32_t* seg,
                          int P, int max_src, int max_tgt, int C, int32_t* row_argmin, int32_t* col_argmin,
                          int32_t* mutual, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Keypoint selection -- replaces np.argsort(scores)[-k:] (test.py:56-57, geometric_registration/evaluate): the k
 * highest-scoring rows of every cloud of a stacked batch, in ascending (score, index) order like a stable argsort's
 * tail.  scores [rows]; seg [P,2] int32 on the device = {offset, length} per cloud; out [P,k] int32 CLOUD-LOCAL row
 * indices; a cloud with fewer than k rows fills its leading k - len slots with -1.  k <= D3F_TOPK_MAX.  One workgroup
 * per cloud: radix select of the k-th largest (score, index) key, then a rank sort of the survivors in LDS.
 * ---------------------------------------------------------------------------------------------- */
#define D3F_TOPK_MAX 6144
int d3f_topk_scores(const float* scores, int rows, const int32_t* seg, int P, int k, int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * KPConv with the non-default influence / aggregation modes -- models/blocks.py:327-352 (KP_influence 'constant' /
 * 'gaussian', aggregation_mode 'closest'; the D3Feat configuration uses 'linear' / 'sum', config.py:39,41, which the
 * fused entry points above implement).  mode = influence (0 linear, 1 constant, 2 gaussian) | 4 for 'closest'.
 * The caller contracts with the kernel weights by plain GEMMs:
 *   forward : wf = aggregate_modes(...);  out = (wf @ W.view(K*Cin, Cout)) / nn
 *   backward: gwf = (grad_out / nn) @ W^T;  grad_W = wf^T (grad_out / nn);  grad_x = grad_input_modes(gwf)
 * wf_out [Nq, K*Cin], nn_out [Nq], gwf [Nq, K*Cin], grad_x [Ns, Cin] (overwritten).  Cin <= 512, K <= 16.
 * ---------------------------------------------------------------------------------------------- */
int d3f_kpconv_aggregate_modes(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                               const float* x, int Cin, const float* kernel_points, int K, float extent, int mode,
                               float* wf_out, float* nn_out, void* stream);
int d3f_kpconv_grad_input_modes(const float* q_pts, int Nq, const float* s_pts, int Ns, const int32_t* idx, int H,
                                int Cin, const float* kernel_points, int K, float extent, int mode, const float* gwf,
                                float* grad_x, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer step with the reference's non-finite-gradient guard -- replaces trainer.py:104-111 (per-parameter
 * torch.isfinite(...).all() host checks, then optimizer.step()) + torch.optim.SGD(momentum, weight_decay) as
 * configured in training_3DMatch.py:62-76, on flat fp32 buffers of n elements (16-byte aligned):
 *   if every grad[i] is finite:  buf = momentum*buf + (grad + weight_decay*params);  params -= lr*buf
 *   else: nothing is modified and state[1] (skipped-step counter) is incremented.
 * state: int32[4] on the device; state[0] is scratch, state[2] / state[3] collect the flags / the number of steps
 * skipped because pair_status (optional, device int32[1]: the status word of the pair the gradient came from --
 * capacity or candidate overflow of its pyramid) was set: such a gradient is treated like a non-finite one.
 * hyper_device: NULL, or float[4] on the device = {lr, momentum, weight_decay, grad_scale}, read when the kernel
 * executes in place of the scalar arguments (grad_scale multiplies the gradient first: 1/world_size turns an
 * all-reduced SUM into the mean without a pass of its own) -- the learning-rate schedule (ExponentialLR, training_3DMatch.py:78-81 stepped at
 * trainer.py:59-60) then changes the step size of an already captured hipGraph.
 * ---------------------------------------------------------------------------------------------- */
int d3f_sgd_guarded_step(const float* grad, float* params, float* momentum_buf, size_t n, float lr, float momentum,
                         float weight_decay, const float* hyper_device, int32_t* state, const int32_t* pair_status,
                         void* stream);
/* The same step on the SUM of n_grads (1..4) gradient buffers -- `grads` is a host array of device pointers, one buffer
 * per pair in flight on this GPU (train.PairLanes: several network steps run concurrently on streams of their own and
 * meet at one optimizer step, the update a data-parallel step over as many ranks makes).  Every buffer is tested for
 * non-finite values on its own; the sum is formed inside the update kernel (no pass of its own). */
int d3f_sgd_guarded_step_lanes(const float* const* grads, int n_grads, float* params, float* momentum_buf, size_t n,
                               float lr, float momentum, float weight_decay, const float* hyper_device, int32_t* state,
                               const int32_t* pair_status, void* stream);
/* data-parallel form of the pair-status gate: poisons grad[0] with NaN before the exchange when the flag is set */
int d3f_poison_gradient_if_status(float* grad, const int32_t* pair_status, int32_t* state, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D3FEAT_HIP_H_ */
