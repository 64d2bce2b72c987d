/* libvnpcc — C-ABI of the MI355X (gfx950) kernels behind the VN-PCN hot path.
 *
 * Drop-in boundary: these entry points are what the reference's Python would bind in place of its two pybind11
 * CUDA extensions and of the eager ATen chains inside models/vn_layers.py.  All functions
 *   - take raw DEVICE pointers, explicit sizes / leading dimensions and the caller's hipStream_t,
 *   - allocate nothing (scratch is passed in), are re-entrant and thread-safe (no mutable global state apart from the
 *     process-wide GEMM-mode switch below, an atomic; per-device kernel attributes are set once per device under an atomic
 *     flag), launch on the given stream without synchronising,
 *   - return 0 on success, VNP_EINVAL (-1) for bad arguments, or the positive hipError_t of a failed launch
 *     (stricter than the reference, which printf()s and swallows launch errors: chamfer3D.cu:145-151).
 *
 * "rows layout": a VN feature map with P points and C channels is a row-major fp32 matrix [P*3, ld] with
 * row = point*3 + t (t = x,y,z) and channels contiguous.  The reference's [B, C, 3, N] tensor
 * (models/vn_layers.py:19) maps to rows[(b*N+n)*3+t][c].  vnp_ref_to_rows / vnp_rows_to_ref convert.
 */
#ifndef VNPCC_H
#define VNPCC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;
#endif

#define VNP_OK 0
#define VNP_EINVAL (-1)

/* ---- Chamfer distance ------------------------------------------------------------------------------------
 * replaces chamfer_3D.forward / chamfer_3D.backward
 *   (/root/reference/extensions/chamfer_distance/chamfer_cuda.cpp:17-33, chamfer3D.cu:136-195).
 * xyz1 [B,n,3], xyz2 [B,m,3] contiguous fp32; dist1 [B,n], dist2 [B,m] fp32; idx1, idx2 int32.
 * ws: scratch of 8*B*max(n,m) bytes.  Outputs need not be pre-zeroed (the reference requires zeroed outputs). */
int vnp_chamfer_fwd(hipStream_t st, int B, int n, const float* xyz1, int m, const float* xyz2,
                    float* dist1, int* idx1, float* dist2, int* idx2, void* ws);
int vnp_chamfer_bwd(hipStream_t st, int B, int n, const float* xyz1, int m, const float* xyz2,
                    const float* grad_dist1, const int* idx1, const float* grad_dist2, const int* idx2,
                    float* grad_xyz1, float* grad_xyz2);
/* Same outputs as vnp_chamfer_fwd, bit for bit, by a pruned exact search for dense clouds (n, m <= 32768): both clouds are
 * Morton-sorted into 64-point chunks with bounding boxes and a wave of 256 queries skips every chunk that cannot hold a
 * nearest neighbour for any of them.  ws: vnp_chamfer_pruned_ws_bytes(B, n, m) bytes (0 = sizes not supported).
 * Diagnostics: after the call the 64-bit word at ws + 8 holds the (query, target) pairs the two scans evaluated (a full scan
 * evaluates 2 B n m): bench.py prices the kernel on this executed work. */
long long vnp_chamfer_pruned_ws_bytes(int B, int n, int m);
/* Diagnostics: byte offset inside ws of the per-wave scanned-chunk counts every call leaves there — unsigned [B][ceil(n / 64)] for
 * direction 1 (queries xyz1), then [B][ceil(m / 64)] for direction 2; a wave = 64 consecutive queries of the Morton-sorted cloud. */
long long vnp_chamfer_pruned_ws_wave_counts(int B, int n, int m);
int vnp_chamfer_fwd_pruned(hipStream_t st, int B, int n, const float* xyz1, int m, const float* xyz2,
                           float* dist1, int* idx1, float* dist2, int* idx2, void* ws);
/* cd_loss_L1 (metrics/loss.py:20-31) on the outputs of vnp_chamfer_fwd: out[3] = {(m1 + m2) / 2, m1, m2} with m = mean sqrt(dist) over ALL
 * n1 = B*n / n2 = B*m elements (double sums in a fixed order; part: 512 doubles of scratch), and its gradient with respect to the two
 * distance arrays, gdist = ((g / 2) / n) / (2 sqrt(dist)) for the device scalar g — the input vnp_chamfer_bwd takes.  Replaces the
 * sqrt / mean / add / div chain of ATen kernels (4 per array forward, 5 backward) by 2 + 1 launches. */
int vnp_cd_l1_loss(hipStream_t st, const float* dist1, long long n1, const float* dist2, long long n2, double* part, float* out);
int vnp_cd_l1_grad(hipStream_t st, const float* g, const float* dist1, long long n1, const float* dist2, long long n2,
                   float* gdist1, float* gdist2);
/* One side of the density-aware Chamfer distance, calc_dcd (/root/reference/metrics/loss.py:77-118; the reference loops over the batch
 * with torch.bincount): hits [B, T] = histogram of the arg-min indices idx [B, N] (scratch, zeroed by the call; integer atomics, so the
 * result does not depend on their order), loss [B] = mean_i (1 - exp(-alpha dist_i) * frac / (hits[idx_i]^lambda + 1e-6)) in the
 * reference's fp32 operation order (the sum over i in double, fixed order), gfac [B, N] = d loss / d dist_i with the weight held
 * constant (the reference detaches it), or NULL. */
int vnp_dcd_side(hipStream_t st, int B, int N, int T, const float* dist, const int* idx, float alpha, float lambda, float frac,
                 int* hits, float* loss, float* gfac);

/* ---- Earth Mover's Distance ---------------------------------------------------------------------------------------
 * replaces emd_cuda.approxmatch_forward / matchcost_forward / matchcost_backward
 *   (/root/reference/extensions/earth_movers_distance/emd.cpp:8-27, emd_kernel.cu:171-193, 257-279, 373-398).
 * match [B,m,n] (zeroed here), temp B*(n+m)*2 floats, ws ceil(m/64)*B floats; outputs are written, not accumulated. */
int vnp_emd_approxmatch(hipStream_t st, int B, int n, int m, const float* xyz1, const float* xyz2, float* match, float* temp);
int vnp_emd_matchcost(hipStream_t st, int B, int n, int m, const float* xyz1, const float* xyz2, const float* match,
                      float* cost, float* ws);
int vnp_emd_matchcost_bwd(hipStream_t st, int B, int n, int m, const float* grad_cost, const float* xyz1, const float* xyz2,
                          const float* match, float* grad1, float* grad2);

/* ---- fp32 GEMM on the matrix cores (MFMA) -----------------------------------------------------------------------
 * replaces the x.transpose(1,-1) -> nn.Linear -> transpose chains of models/vn_layers.py:21,36,65,69,162 and their
 * autograd.  C[M,N] (+)= op(A) op(B) (+ bias).
 *   op_a = 0: A is [M,K] row-major; 1: A is [K,M] row-major.   op_b = 0: B is [N,K] row-major; 1: B is [K,N].
 *   bias: optional [groups*3, ldbias]; row r gets bias[(r / rows_per_group)*3 + r%3] (per-sample VN bias).
 *   accumulate != 0: C += result.   nsplit > 1: split-K through ws (nsplit*M*N floats), bias not allowed. */
/* GEMM arithmetic: 0 (default) = every fp32 operand split exactly into three bf16 terms, six bf16 MFMA products with fp32
 * accumulation (error vs fp64 at or below mode 1's); 1 = fp32-input MFMA.  Process-wide; returns 0 or VNP_EINVAL. */
int vnp_gemm_set_mode(int mode);
int vnp_gemm_get_mode(void);
/* The tile vnp_gemm runs for an aligned split-mode product with A [M, K] (op_a = 0; N % 128 == 0, K % 32 == 0): 192 / 128 / 96 = the
 * row-tile height of the 256-thread kernel, 256 = the 512-thread 192 x 256 kernel.  Chosen per launch from the number of workgroup
 * rounds each height needs (measured costs per round, gemm.hip: vnp_choose_tile); the results are the same bits whichever runs. */
int vnp_gemm_tile_rows(int op_b, int M, int N, int nsplit, int nbatch, int has_bias);
/* Measurement hook (bench.py, no reference counterpart): the NEXT matrix-product launch of the calling host thread (vnp_gemm*,
 * vnp_gemm_vn_stats*, vnp_linear_pool*_fwd) stamps e0 / e1 with the start and the end of its ONE dominant kernel's dispatch on the
 * launch stream (hipExtLaunchKernelGGL: no marker packets; not the slab reduction or the pool's candidate passes) and stores in *kind (host memory, may be NULL) what that kernel was:
 * 0 six-product bf16 split kernel, 1 fp32-input MFMA kernel, 2 streaming kernel of a degenerate shape (K = 1, N = 1, M = 1: no
 * MFMA), 3 screening pass with one fp16 product, 4 screening pass with three bf16 products, 5 two fp16 planes / three products. */
int vnp_gemm_profile_next(hipEvent_t e0, hipEvent_t e1, int* kind);
/* Drop a pending request.  (Every matrix-product entry point also drops a request it did not consume when it returns — argument error,
 * a path that brackets no kernel — so a request never outlives the call it was armed for.) */
int vnp_gemm_profile_cancel(void);
int vnp_gemm(hipStream_t st, int op_a, int op_b, int M, int N, int K,
             const float* A, int lda, const float* B, int ldb, float* C, int ldc,
             const float* bias, int ldbias, int rows_per_group,
             int accumulate, int nsplit, float* ws);
/* C = op(A) op(B) + R: vnp_gemm (no bias) with a residual connection taken in the epilogue — models/pointr/vn_pointr.py:401-410,
 * models/transformer.py:60-65 add a block's attention / MLP branch to its input right after the branch's last VNLinear.  R [M][N] with
 * row stride ldr >= N; R may be C.  With split-K the slab reduction adds R. */
int vnp_gemm_add(hipStream_t st, int op_a, int op_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                 float* C, int ldc, const float* R, int ldr, int nsplit, float* ws);
/* vnp_gemm given the operands' absolute maxima (opt-in fast path): amax_a / amax_b (both or neither) are device words holding the bit
 * pattern of max |A| and max |B| over the WHOLE operand (vnp_absmax; any upper bound within a few binades works, an element ABOVE the
 * word overflows to Inf / NaN).  Aligned shapes with K >= 64 then run on two fp16 planes of the power-of-two scaled operands and three
 * MFMA products instead of three bf16 planes and six; the partial sum of every 32-wide K-tile is added to the running sum in fp32 with
 * round-to-nearest.  Accuracy: operands are kept to 22-23 bits relative to themselves down to 2^-16 of the operand's largest element
 * and to 2^-39 of that element below — normwise at or below an fp32 FMA-chain sgemm against fp64 for K >= 64, not element by element.
 * Other shapes, and null words: exactly vnp_gemm. */
int vnp_gemm_ex(hipStream_t st, int op_a, int op_b, int M, int N, int K,
                const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, int ldbias, int rows_per_group,
                int accumulate, int nsplit, float* ws, const unsigned* amax_a, const unsigned* amax_b);
/* out[0] = bit pattern of max |x[r][c]| over rows x cols (x 16-byte aligned, cols % 4 == 0, ld % 4 == 0); NaN ranks above Inf */
int vnp_absmax(hipStream_t st, const float* x, long long ld, long long rows, int cols, unsigned* out);

/* nbatch independent products C_i = op(A_i) op(B_i), element strides bsa/bsb/bsc (attention: one per (sample, head)) */
/* vnp_gemm (NT, optional per-sample VN bias) that also returns, for the first stats_cols columns, the per-1024-point Welford
 * partials vnp_vn_norm_stats would compute from C — taken from the accumulators, so the BatchNorm statistics of a VN layer cost
 * no extra pass over its output.  part1024 [M/3072][stats_cols][2]; ws: vnp_gemm_vn_stats_ws_floats(M, stats_cols) floats.
 * VNP_EINVAL unless split mode, M % 3072 == 0, N % 128 == 0, K % 32 == 0, stats_cols % 64 == 0, aligned operands. */
long long vnp_gemm_vn_stats_ws_floats(long long M, int stats_cols);
int vnp_gemm_vn_stats(hipStream_t st, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                      const float* bias, int ldbias, int rows_per_group, int stats_cols, float* part1024, float* ws);
/* the same with the operands' absolute maxima (see vnp_gemm_ex) */
int vnp_gemm_vn_stats_ex(hipStream_t st, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                      const float* bias, int ldbias, int rows_per_group, int stats_cols, float* part1024, float* ws,
                         const unsigned* amax_a, const unsigned* amax_b);
int vnp_gemm_batched(hipStream_t st, int op_a, int op_b, int M, int N, int K,
                     const float* A, int lda, long long bsa, const float* B, int ldb, long long bsb,
                     float* C, int ldc, long long bsc, int nbatch, int accumulate);

/* ---- VN transformer pieces (models/transformer.py:25-106, models/vn_layers.py:129-150) ------------------------------
 * VNLayerNorm: LayerNorm over the channel axis of the vector norms of each point; stats [P][2] = (mean, rstd) */
int vnp_vn_layernorm_fwd(hipStream_t st, const float* x, int ldx, long long P, int C, const float* gamma, const float* beta,
                         float* out, int ldo, float* stats);
/* part [ceil(P/4)][2][C] partial (dgamma | dbeta) sums, reduce over the first axis with vnp_reduce_rows */
int vnp_vn_layernorm_bwd(hipStream_t st, const float* go, int ldgo, const float* x, int ldx, long long P, int C,
                         const float* gamma, const float* beta, const float* stats, float* gx, int ldgx, float* part);
/* VNLayerNorm of s = x + a1 (+ a2, may be NULL), with s written to sum_out [P*3][lds] (the residual stream a transformer block carries:
 * x + pos, models/pointr/vn_pointr.py:700, and vn_x + x_2 of the block before, :410); sum_out must not alias an input.
 * _bwd_add: the backward of VNLayerNorm plus gres, the gradient that reaches the same tensor through the block's skip connection. */
int vnp_vn_add_layernorm_fwd(hipStream_t st, const float* x, int ldx, const float* a1, int lda1, const float* a2, int lda2,
                             long long P, int C, const float* gamma, const float* beta, float* sum_out, int lds,
                             float* out, int ldo, float* stats);
int vnp_vn_layernorm_bwd_add(hipStream_t st, const float* go, int ldgo, const float* x, int ldx, long long P, int C,
                             const float* gamma, const float* beta, const float* stats, const float* gres, int ldgres,
                             float* gx, int ldgx, float* part);
/* rows [B*N*3][ld] <-> heads [B][H][N][(C/H)*3] (transformer.py:91-93: reshape(B,H,C/H,3,N).permute(0,1,4,2,3)) */
int vnp_rows_to_heads(hipStream_t st, const float* rows, int ld, int B, int N, int C, int H, float* heads);
int vnp_heads_to_rows(hipStream_t st, const float* heads, int B, int N, int C, int H, float* rows, int ld);
/* the same for G matrices side by side in the rows (q | k | v of ONE projection GEMM: rows [B*N*3][G*C], any row stride ld >= G*C):
 * heads = G tensors [B*H][N][3 C / H], one after the other — the three permutes of an attention (forward), or of its gradient, in one
 * launch.  G*C <= 5000. */
int vnp_rows_to_heads_grouped(hipStream_t st, const float* rows, int ld, int B, int N, int C, int H, int G, float* heads);
int vnp_heads_to_rows_grouped(hipStream_t st, const float* heads, int B, int N, int C, int H, int G, float* rows, int ld);
/* The same permutation into layouts the aligned GEMM kernels can take when the head dimension Dh = 3 C / H (144 in
 * Attention_VN_FoldingNet) is neither a multiple of 32 nor of 128:
 *   _pad: token-major heads [(b, h)][n][ldh], ldh >= Dh, columns [Dh, ldh) zero   (K of q k^T and dout v^T, padded to a multiple of 32)
 *   _fm:  feature-major heads [(b, h)][fr][N], fr >= Dh, rows [Dh, fr) zero        (the head dimension on M of out^T = v^T P^T,
 *         dv^T = dout^T P, dq^T = k^T dS^T, dk^T = q^T dS, padded to whole row tiles) and back to rows. */
int vnp_rows_to_heads_pad(hipStream_t st, const float* rows, int ld, int B, int N, int C, int H, float* heads, int ldh);
int vnp_rows_to_heads_fm(hipStream_t st, const float* rows, int ld, int B, int N, int C, int H, float* heads, int fr);
int vnp_heads_fm_to_rows(hipStream_t st, const float* heads, int fr, int B, int N, int C, int H, float* rows, int ld);
/* S[R][L] <- softmax(S*scale) in place; backward: dP <- P*(dP - sum(dP*P))*scale */
int vnp_softmax_fwd(hipStream_t st, float* S, long long R, int L, float scale);
int vnp_softmax_bwd(hipStream_t st, const float* P, float* dP, long long R, int L, float scale);

/* ---- VN BatchNorm / leaky projection ------------------------------------------------------------------------
 * VNBatchNorm (models/vn_layers.py:107-127), VNLeakyReLU (:25-43), VNLinearLeakyReLU tail (:66-73).
 * Chunked reductions use chunks of vnp_chunk_points() points: nchunk = ceil(P / vnp_chunk_points()).  For the per-channel
 * passes over a [P, C] map vnp_chunk_points_pc(P, C) gives the chunk that fills the chip (1024 halved down to 16 until the grid has
 * >= 1024 workgroups); vnp_vn_norm_stats / vnp_vn_bn_leaky_bwd1 take it as `chunk_points` (0 = vnp_chunk_points()) and write one
 * partial row per chunk of that size; vnp_vn_bn_finalize_chunked merges statistics partials of any chunk size. */
int vnp_chunk_points(void);
int vnp_chunk_points_pc(long long P, int C);
/* partial [nchunk][C][2] = per-chunk (mean, M2 = sum (n - mean)^2) of n = |p| + 1e-6 (Welford; merged by Chan) */
int vnp_vn_norm_stats(hipStream_t st, const float* p, int ldp, long long P, int C, float* partial, int chunk_points);
/* mean/rstd [C] from the partials; running stats (nullable) updated like torch BatchNorm (momentum, unbiased var) */
int vnp_vn_bn_finalize(hipStream_t st, const float* partial, int nchunk, int C, long long count,
                       float* mean, float* rstd, float* running_mean, float* running_var, float momentum, float eps);
int vnp_rstd_from_var(hipStream_t st, const float* var, float* rstd, int C, float eps);
/* out = leaky(bn(p), d); d NULL: BatchNorm only; mean NULL: no BatchNorm */
int vnp_vn_bn_leaky_fwd(hipStream_t st, const float* p, int ldp, const float* d, int ldd, float* out, int ldo,
                        long long P, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        float negative_slope);
/* backward phase 1: gq = dL/dbn(p), gd = dL/dd, partial [nchunk][C][2] = (sum g_nhat, sum g_nhat*xhat).
 * go NULL: rank-1 upstream gradient go[r][c] = go_row[r]*go_col[c]. */
int vnp_vn_bn_leaky_bwd1(hipStream_t st, const float* go, int ldgo, const float* go_row, const float* go_col,
                         const float* p, int ldp, const float* d, int ldd,
                         float* gq, int ldgq, float* gd, int ldgd, long long P, int C,
                         const float* mean, const float* rstd, const float* gamma, const float* beta,
                         float negative_slope, float* partial, float* gw_partial, int chunk_points);
/* The same backward in two read-only-heavy passes that never write dL/dq (6 maps instead of 7 in rank-1 form):
 * pass 1: partial (+ gw_partial) only;  pass 2: gq = dL/dp — phase 2 applied on the fly with the reduced sums S [C][2]
 * (NULL: eval mode) — and gd; optionally colsum [nchunk][3][2C] = per-chunk sums over points of (dL/dp | dL/dd) per component (the
 * per-sample bias gradient of a layer with de-duplicated broadcast channels).  BatchNorm is mandatory here (mean non-NULL). */
int vnp_vn_bn_leaky_bwd_pass(hipStream_t st, int pass, const float* go, int ldgo, const float* go_row, const float* go_col,
                             const float* p, int ldp, const float* d, int ldd,
                             float* gq, int ldgq, float* gd, int ldgd, long long P, int C,
                             const float* mean, const float* rstd, const float* gamma, const float* beta,
                             float negative_slope, float* partial, float* gw_partial, int chunk_points,
                             const float* S, long long count, float* colsum);
/* gw_partial (rank-1 mode only, nullable) [nchunk][C]: per-chunk sums of go_row . out, i.e. dL/dw of a final VNLinear(C->1)
 * whose input (this layer's output) was never stored.
 * Forward of that fusion: fine[pt*3+t] = sum_c leaky(bn(p),d)[pt,t,c]*w[c] + add[(pt/add_group)*3+t] (models/pcn.py:341-346,387) */
int vnp_vn_bn_leaky_rowdot_fwd(hipStream_t st, const float* p, int ldp, const float* d, int ldd, long long P, int C,
                               const float* mean, const float* rstd, const float* gamma, const float* beta,
                               float negative_slope, const float* w, const float* add, int add_group, float* out);
/* backward phase 2: g (dL/dbn(p)) -> dL/dp in place; S [C][2] reduced partial sums (NULL: eval-mode BatchNorm) */
int vnp_vn_bn_bwd2(hipStream_t st, float* g, int ldg, const float* p, int ldp, long long P, int C,
                   const float* mean, const float* rstd, const float* gamma, const float* beta,
                   const float* S, long long count);
/* out[b][j] = sum_{i<n} in[b*batch_stride + i*stride + j], j < W (double accumulation) */
int vnp_reduce_rows(hipStream_t st, const float* in, int nb, long long batch_stride, int n, long long stride, int W,
                    float* out, long long out_stride);
/* Sum over n partials of interleaved pairs: in [n][C][2] -> S [C][2] (read by the BatchNorm backward kernels) and planar [2][C]
 * (the two per-channel parameter gradients dL/dbeta, dL/dgamma as contiguous rows). */
int vnp_reduce_pairs(hipStream_t st, const float* in, int n, int C, float* S, float* planar);

/* ---- pooling ----------------------------------------------------------------------------------------------
 * VNMaxPool (models/vn_layers.py:153-167) over groups of gs consecutive points; idx [G][C] = local arg-max (first max).
 * ws_val / ws_idx: G*ceil(gs/vnp_chunk_points())*C elements. */
/* VNLinear followed by VNMaxPool in ONE pass (models/pcn.py:173-174): the pooled x = W f and the direction d = W' f
 * (W' = Wd W) are produced by one split-GEMM whose epilogue forms <x, d> and the arg-max in registers; neither matrix is
 * written.  Wcat [2C, K]: per 32 channels, 32 rows of W then 32 rows of W'.  ws: vnp_linear_pool_ws_bytes(G, gs, C).
 * VNP_EINVAL unless split GEMM mode, (G*gs*3) % 192 == 0, gs % 32 == 0, C % 64 == 0, K % 32 == 0 and 16-byte alignment. */
long long vnp_linear_pool_ws_bytes(long long G, int gs, int C);
int vnp_linear_pool_fwd(hipStream_t st, const float* x, int ldx, const float* Wcat, int ldw, long long G, int gs, int C,
                        int K, int* idx, float* out, int ldo, void* ws);
/* As vnp_linear_pool_fwd when the linear output y = x W^T has another consumer as well (encoder first_conv[1] -> maxpool1 and
 * second_conv[0], models/pcn.py:169-172): y [G*gs*3, ldy >= C] is stored too; the direction still never exists in memory. */
int vnp_linear_pool_store_fwd(hipStream_t st, const float* x, int ldx, const float* Wcat, int ldw, long long G, int gs, int C,
                              int K, float* y, int ldy, int* idx, float* out, int ldo, void* ws);
/* The same operator — same arguments, same restrictions, same results BIT FOR BIT — with an exact screened arg-max: a 3-product
 * pass (half the MFMA work) brackets every score between rigorous bounds, the ~1-2 surviving candidates per (group, channel) are
 * evaluated exactly as diagonals of MFMA tiles with the full kernels' product sequence, and a predicated launch of the full
 * kernel takes over if the candidate list (32 pairs per (group, channel) on average) overflows or a bound is not finite (NaN / Inf
 * inputs, massive exact ties).
 * ws: vnp_linear_pool_screened_ws_bytes(G, gs, C, K) bytes; afterwards its first three ints are {candidates evaluated,
 * fallback flag, candidate capacity} (diagnostics). */
/* flags: 0 = bracketing pass on three bf16 products (truncation 2^-16); VNP_POOL_SCREEN_FP16 = on ONE fp16 product (2^-10: about twice
 * the candidates, 60 % less pass time; operands beyond fp16's range raise the fallback flag; needs K % 64 == 0, else the three-product
 * pass runs).  The fp16 pass reads fp16 images of x and Wcat that the row-norm pass writes into ws.  Either way the result is exact. */
#define VNP_POOL_SCREEN_FP16 1
/* The fp16 pass picks its tile from the shape: 384 x 256 on LDS-DMA staging (M % 384 == 0, 2C % 256 == 0: screen_xwide_kernel.h),
 * else 192 x 256 (2C % 256 == 0), else 192 x 128; all three write the same slots bit for bit.  These two flags only narrow the choice
 * (tests hold the tiles against each other): no 384-row tile / no 256-column tile either. */
#define VNP_POOL_SCREEN_NO_XWIDE 2
#define VNP_POOL_SCREEN_NARROW 4
long long vnp_linear_pool_screened_ws_bytes(long long G, int gs, int C, int K);
int vnp_linear_pool_screened_fwd(hipStream_t st, const float* x, int ldx, const float* Wcat, int ldw, long long G, int gs, int C,
                                 int K, int* idx, float* out, int ldo, void* ws, int flags);
int vnp_vn_pool_fwd(hipStream_t st, const float* x, int ldx, const float* d, int ldd, long long G, int gs, int C,
                    int* idx, float* out, int ldo, float* ws_val, int* ws_idx);
int vnp_vn_pool_bwd(hipStream_t st, const float* gout, int ldgo, const int* idx, long long G, int gs, int C,
                    float* gx, int ldgx);
/* same scatter ADDED into an existing gx (not zero-filled): the pooled tensor's other consumer has already written its gradient
 * there (encoder first_conv[1] feeds maxpool1 and second_conv[0], models/pcn.py:169-172) */
int vnp_vn_pool_bwd_acc(hipStream_t st, const float* gout, int ldgo, const int* idx, long long G, int gs, int C,
                        float* gx, int ldgx);
/* Stable sort of every group's arg-max indices (what vnp_pool_lin_bwd / vnp_dir_pool_bwd take as `pts` / `order`; replaces
 * torch.sort(idx, dim=1, stable=True) and its dtype conversions): idx [G][C] int32 with values in [0, gs) -> pts [G][C] ascending,
 * order [G][C] the positions they came from, ties in position order.  C <= 4096. */
int vnp_sort_groups(hipStream_t st, const int* idx, long long G, int C, int gs, int* pts, int* order);
/* Backward of VNLinear followed by VNMaxPool (models/pcn.py:169-170,173-174) without the dense gradient of the linear output:
 * gW[Co][Ci] = sum_{g,t} gout[g*3+t][o] x[(g*gs+idx[g][o])*3+t][:]   (NULL: skip)
 * gx[G*gs*3][Ci] (every row written, zeros for a point no channel selected: no fill beforehand) gets, for every selected point, sum_{o: idx[g][o]==n} gout[g*3+t][o] W[o][:]   (NULL: skip)
 * order/pts [G][Co]: channels of each group sorted by arg-max point, and those points (ascending). */
int vnp_pool_lin_bwd(hipStream_t st, const float* gout, int ldgo, const int* idx, const int* order, const int* pts,
                     const float* x, int ldx, const float* W, int ldw, int G, int gs, int Co, int Ci,
                     float* gW, int ldgw, float* gx, int ldgx);
/* mean_pool (models/vn_layers.py:170-171) over groups of gs consecutive points */
int vnp_vn_mean_pool_fwd(hipStream_t st, const float* x, int ldx, long long G, int gs, int C, float* out, int ldo);
int vnp_vn_mean_pool_bwd(hipStream_t st, const float* gout, int ldgo, long long G, int gs, int C, float* gx, int ldgx);

/* ---- FoldingNet expansion (models/pcn.py:375-387, broadcast input de-duplicated) ---------------------------
 * out[((b*Mc+mc)*S+s)*3+t][j] = bias[b*3+t][j] + wseed[j]*seed[(b)][s][t] + wcoarse[j]*coarse[b][mc][t], j < W */
int vnp_fold_fwd(hipStream_t st, const float* bias, int ldb, const float* wseed, const float* wcoarse,
                 const float* seed, int seed_batched, const float* coarse, int B, int Mc, int S, int W,
                 float* out, int ldo);
/* part [B*Mc][5][W]: rows 0-2 sum_s g (per t), row 3 sum g*seed, row 4 sum g*coarse */
int vnp_fold_bwd(hipStream_t st, const float* g, int ldg, const float* seed, int seed_batched, const float* coarse,
                 int B, int Mc, int S, int W, float* part);

/* ---- graph construction / gathers: replace KNN_CUDA 0.2 and pointnet2_ops (third-party CUDA packages, README.md:29-31;
 * call sites models/dgcnn.py:11,210-218,236,257-259, models/pcn.py:6-9).  Point-major clouds [B,N,D].
 * knn: k nearest (squared L2) of each query among ref, sorted by (distance, index); idx [B][Nq][k] local indices. */
int vnp_knn(hipStream_t st, int B, int D, const float* ref, int Nr, const float* qry, int Nq, int k, int* idx, float* dist);
/* same contract and bits as vnp_knn for large D (feature-space graphs): distance matrix once (ws: B*Nq*Nr floats), then a
 * per-row selection; Nr <= 16384 */
int vnp_knn_matrix(hipStream_t st, int B, int D, const float* ref, int Nr, const float* qry, int Nq, int k, int* idx, float* dist,
                   float* ws);
/* furthest_point_sample: starts at index 0, ties to the lowest index; idx [B][M]; N <= 16384.  vnp_fps: plain FPS, every
 * point is a candidate.  vnp_fps_ex(flags = VNP_FPS_SKIP_NEAR_ORIGIN): the convention of pointnet2_ops (the package the
 * reference calls, models/pcn.py:7, models/dgcnn.py:210; its furthest_point_sampling_kernel has `if (mag <= 1e-3) continue;`):
 * a candidate with x^2+y^2+z^2 <= 1e-3 is never selected and its running distance is not updated (point 0 is still sample 0;
 * if every candidate is skipped the pick is index 0).  The Python binding uses the pointnet2_ops convention by default. */
#define VNP_FPS_SKIP_NEAR_ORIGIN 1
int vnp_fps(hipStream_t st, int B, int N, const float* xyz, int M, int* idx);
int vnp_fps_ex(hipStream_t st, int B, int N, const float* xyz, int M, int* idx, int flags);
/* what follows every furthest_point_sample in the reference (gather_operation on the coordinates, models/dgcnn.py:210-218, and the
 * `idx + arange(B) * N` that turns local picks into row ids): out [B][M][3] = xyz[b][idx[b][i]], gid [B*M] = b * N + idx[b][i] (gid may
 * be NULL), one launch. */
int vnp_gather_coords(hipStream_t st, int B, int N, const float* xyz, const int* idx, int M, float* out, int* gid);
/* rows layout: out point i <- x point src[i] (gather_operation); and its adjoint gx[s] = sum over the i with src[i] == s of g[i]:
 * a gather over the reverse adjacency of src (vnp_edge_reverse), summed in ascending i in double — no atomics, every row of gx written,
 * the same bits on every run.  ws: 2 * (Pin + 1) + 2 * Pout ints. */
int vnp_gather_points(hipStream_t st, const float* x, int ldx, const int* src, long long Pout, int W, float* out, int ldo);
int vnp_scatter_add_points(hipStream_t st, const float* g, int ldg, const int* src, long long Pout, int W,
                           float* gx, int ldgx, long long Pin, int* ws);
/* Fused EdgeConv: VNLinearLeakyReLU on the kNN edge tensor + mean over the k neighbours (models/dgcnn.py:250-278 and the
 * conv1/4/5/6 blocks; VN-PoinTr grouper) without materialising the edge tensor: (p | d)(q, j) = u[nbr[q][j]] + v[q] is recomputed
 * in each pass from the per-point matrices u = x W1^T, v = x (W2 - W1)^T (columns [feat | dir]).
 * mode 0: Welford partials of |p| per block of vnp_edge_block_queries() queries -> vnp_vn_bn_finalize_chunked(chunk = 64 k);
 * mode 1: out[q] = mean_j leaky(bn(p), d); mode 2: backward sums (dL/dbeta, dL/dgamma partials); mode 3: gv[q] = sum_j g(q, j),
 * gu / ldgu / Pu must be NULL / 0 / 0 (kept in the signature; the scatter form with float atomics is gone).  count = number of edges.
 * The source points' gradient: vnp_edge_fused_gather — gu[s] = sum over the edges INTO s (reverse adjacency from vnp_edge_reverse:
 * start [Pu + 1], rev [Pq * k] query ids grouped by source and ASCENDING inside a source's list; ws (Pu + 1) + Pq * k ints), the edge
 * gradient recomputed term for term and summed in double in list order: no atomics, no zero fill, the same bits on every run. */
int vnp_edge_block_queries(void);
int vnp_vn_bn_finalize_chunked(hipStream_t st, const float* partial, int nchunk, int chunk_points, int C, long long count,
                               float* mean, float* rstd, float* running_mean, float* running_var, float momentum, float eps);
int vnp_edge_fused(hipStream_t st, int mode, const float* u, int ldu, const float* v, int ldv, const int* nbr,
                   long long Pq, int k, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                   float negative_slope, float* partial, float* out, int ldo, const float* go, int ldgo,
                   const float* sums, long long count, float* gu, int ldgu, long long Pu, float* gv, int ldgv);
int vnp_edge_reverse(hipStream_t st, const int* nbr, long long Pq, int k, long long Pu, int* start, int* rev, int* ws);
int vnp_edge_fused_gather(hipStream_t st, const float* u, int ldu, const float* v, int ldv, const int* start, const int* rev,
                          long long Pu, int k, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                          float negative_slope, const float* go, int ldgo, const float* sums, long long count, float* gu, int ldgu);
/* vnp_edge_fused / vnp_edge_fused_gather for a layer whose input is ONE VN channel — the points themselves: conv1 of VN_DGCNN_fps and of
 * the VN-PoinTr grouper (models/dgcnn.py:291, models/pointr/utils/dgcnn_group.py:225).  xu [Pu*3], xv [Pq*3]: coordinates; wu, wv [2C]:
 * the single column of [Wu; Wv] (vnp_edge_weights).  u = x wu^T and v = x wv^T are never formed: an edge reads its neighbour's 12 bytes
 * instead of 24 C, and the product that formed them is gone.  Forward passes: the same bits as the general form on the formed matrices;
 * backward: the same arithmetic, equal to rounding (the compiler may contract a product and a sum differently in the two instantiations). */
int vnp_edge_fused_r1(hipStream_t st, int mode, const float* xu, const float* xv, const float* wu, const float* wv, const int* nbr,
                      long long Pq, int k, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                      float negative_slope, float* partial, float* out, int ldo, const float* go, int ldgo,
                      const float* sums, long long count, float* gv, int ldgv);
int vnp_edge_fused_gather_r1(hipStream_t st, const float* xu, const float* xv, const float* wu, const float* wv, const int* start,
                             const int* rev, long long Pu, int k, int C, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, float negative_slope, const float* go, int ldgo, const float* sums, long long count,
                             float* gu, int ldgu);
/* The per-point weights [Wu; Wv] [4C'][C] of a fused EdgeConv layer from its VNLinearLeakyReLU weights Wf, Wd [C'][2C] (contiguous):
 * Wu = [Wf[:, :C]; Wd[:, :C]] acts on the neighbour, Wv = [Wf[:, C:] - Wf[:, :C]; Wd[:, C:] - Wd[:, :C]] on the centre
 * (W cat(f_j - x, x) = W1 f_j + (W2 - W1) x, models/dgcnn.py:250-278).  dir = 0: uv <- (Wf, Wd); dir = 1: the adjoint, (Wf, Wd) <- uv. */
int vnp_edge_weights(hipStream_t st, int dir, float* Wf, float* Wd, float* uv, int Cp, int C);
/* out[(p*k+j)*3+t][c] = u[nbr[p*k+j]*3+t][c] + v[p*3+t][c]: edge tensor of vn_get_graph_feature (models/dgcnn.py:250-278)
 * with the layer's weights applied per point first (W [feat-x | x] = W1 feat + (W2-W1) x) */
int vnp_edge_gather_add(hipStream_t st, const float* u, int ldu, const float* v, int ldv, const int* nbr,
                        long long P, int k, int W, float* out, int ldo);

/* Fused FoldingNet layer 0 (rank-2 expansion + BatchNorm + leaky projection) that never writes the [P*3, 2C] (p | d)
 * tensor: p and d are recomputed from bias / wseed / wcoarse / seed / coarse in every pass.
 *   mode 0: partial [nchunk][C][2] = (mean, M2) of |p|+eps           (then vnp_vn_bn_finalize)
 *   mode 1: out [P*3][ldo] = leaky(bn(p), d)
 *   mode 2: partial [nchunk][C][2] = (sum g_nhat, sum g_nhat*xhat) from the upstream gradient go
 *   mode 3: part5 [nchunk][5][2C] = per-chunk sums for dbias (rows 0-2), dwseed (3), dwcoarse (4); gcoarse [ceil(C / 64)][B*Mc][3]
 *           = dL/dcoarse as one partial per 64-column block (plain stores; the caller sums the first axis: no atomics, the same bits
 *           on every run); sums [C][2] = reduced mode-2 partials (NULL: eval-mode BatchNorm)
 * nchunk = B*Mc / coarse_per_chunk; coarse_per_chunk divides Mc and coarse_per_chunk*S <= vnp_chunk_points() (0: vnp_chunk_points()/S):
 * a chunk is coarse_per_chunk consecutive coarse points with their S dense points each and never straddles two samples. */
int vnp_fold_fused(hipStream_t st, int mode, const float* bias, int ldb, const float* wseed, const float* wcoarse,
                   const float* seed, int seed_batched, const float* coarse, int B, int Mc, int S, int C,
                   const float* mean, const float* rstd, const float* gamma, const float* beta, float negative_slope,
                   float* out, int ldo, const float* go, int ldgo, float* partial, const float* sums, long long count,
                   float* part5, float* gcoarse, int coarse_per_chunk);
/* The four passes of vnp_fold_fused for a layer whose per-point input is ONE VN channel x [G][S][3] beside per-GROUP features
 * (Attention_VN_FoldingNet's folding stages, /root/reference/models/pcn.py:486-488: cat([point, centre features]) ->
 * VNLinearLeakyReLU(385, 256)):  p | d [(g, s)][t][j] = bias[g*3 + t][j | C + j] + w[j | C + j] * x[g][s][t],  bias [G*3][2C] = the
 * group features times their weight columns, w [2C] = the weight column of the point channel.  The [G*S*3, 2C] pre-activations are
 * never written.  mode 0: Welford partials [G/cpc][C][2]; 1: out [G*S*3][C]; 2: backward sums partial [G/cpc][C][2]; 3: gbias
 * [G*3][2C] (per group), part5 [G/cpc][5][2C] (row 3 = dL/dw partials; rows 0-2, 4 unused) and, if gx_part != NULL, dL/dx as one
 * partial per 64-column block, gx_part [ceil(C/64)][G][S*3] (S <= 16; the caller sums the blocks).  S even for modes 2 and 3;
 * coarse_per_chunk groups per chunk (divides G; 0 = vnp_chunk_points() / S). */
int vnp_group_fold_fused(hipStream_t st, int mode, const float* bias, int ldb, const float* w, const float* x, long long G, int S, int C,
                         const float* mean, const float* rstd, const float* gamma, const float* beta, float negative_slope,
                         float* out, int ldo, const float* go, int ldgo, float* partial, const float* sums, long long count,
                         float* part5, float* gbias, float* gx_part, int coarse_per_chunk);

/* ---- non-equivariant baselines ("foldingnet" decoder, "dgcnn_fps" encoder: /root/reference/models/pcn.py:274-317, dgcnn.py:19-161) ----
 * nn.BatchNorm1d over the point-major rows x [P, C] of a [B, C, N] tensor together with the activation behind it (act: 0 none, 1 ReLU,
 * 2 LeakyReLU(slope)).  mode 0: Welford partials [ceil(P / vnp_bn1d_chunk_rows())][C][2] (merge with vnp_vn_bn_finalize_chunked, which
 * also updates the running statistics);  1: y = act((x - mean) rstd gamma + beta);  2: backward sums partials of the same shape
 * (reduce with vnp_reduce_pairs: S [C][2] = (dL/dbeta, dL/dgamma));  3: y <- dL/dx from gy (sums = S, or NULL in eval mode). */
int vnp_bn1d_chunk_rows(void);
int vnp_bn1d_act(hipStream_t st, int mode, const float* x, int ldx, long long P, int C, const float* mean, const float* rstd,
                 const float* gamma, const float* beta, int act, float slope, float* y, int ldy, const float* gy, int ldgy,
                 float* partial, const float* sums, long long count);

/* EdgeConv block of the non-equivariant DGCNN_fps (/root/reference/models/dgcnn.py:101-124: Conv2d(1x1) on cat(x_k[nbr] - x_q, x_q) ->
 * GroupNorm(G) -> LeakyReLU(slope) -> max over the k neighbours) with the convolution applied per point first (u [Pk][C] = W1 x_k,
 * v [Pq][C] = (W2 - W1) x_q; e(q, j) = u[nbr[q][j]] + v[q]) and the edge tensor never written.  B samples of nq consecutive queries;
 * nbr [B*nq][k] global row ids of u; G divides C; k <= 255.
 *   mode 0: partial [B * ceil(nq / 64)][C][2] = Welford (mean, M2) per (block of 64 queries, channel) -> vnp_edge_gn_finalize -> mean, rstd [B][G]
 *   mode 1: out [Pq][ldo] = max_j leaky(gn(e)), amax [Pq][C] = the j attaining it
 *   mode 2: partial [B * ceil(nq / 64)][C][4] = (dbeta, dgamma, s1, s2) column sums per block; the caller reduces them per sample -> sums [B][C][4]
 *   mode 3: gv [Pq][ldgv] = dL/dv;  vnp_edge_gn_gather: gu [Pk][ldgu] = dL/du as a gather over the reverse adjacency (vnp_edge_reverse):
 *           no atomics, the same bits on every run. */
int vnp_edge_gn_max(hipStream_t st, int mode, const float* u, int ldu, const float* v, int ldv, const int* nbr, int B, int nq, int k,
                    int C, int G, const float* mean, const float* rstd, const float* gamma, const float* beta, float slope,
                    float* out, int ldo, unsigned char* amax, const float* go, int ldgo, float* partial, const float* sums,
                    float* gv, int ldgv);
int vnp_edge_gn_finalize(hipStream_t st, const float* partial, int B, int nq, int k, int C, int G, float eps, float* mean, float* rstd);
int vnp_edge_gn_gather(hipStream_t st, const float* u, int ldu, const float* v, int ldv, const int* nbr, const int* start,
                       const int* rev, long long Pk, int B, int nq, int k, int C, int G, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, float slope, const unsigned char* amax, const float* go, int ldgo,
                       const float* sums, float* gu, int ldgu);
/* y = act(x + bias) on rows [P][C] (bias [C] or NULL; act 0 none / 1 ReLU / 2 LeakyReLU(slope)); with gy / gx: gx = gy act'(x + bias)
 * (nn.Linear / Conv1d biases and the ReLU of the baselines' heads: models/dgcnn.py:48-54, models/pcn.py:281-289) */
int vnp_bias_act(hipStream_t st, const float* x, int ldx, const float* bias, long long P, int C, int act, float slope,
                 float* y, int ldy, const float* gy, int ldgy, float* gx, int ldgx);
/* out [B][C] = max over the n rows of each sample of x [B*n][ldx] (torch.max(dim) of models/dgcnn.py:157), amax [B][C] the first row
 * attaining it; backward: gx [B*n][C] = go at that row, 0 elsewhere (every element written) */
int vnp_rows_max(hipStream_t st, const float* x, int ldx, int B, int n, int C, float* out, int* amax);
int vnp_rows_max_bwd(hipStream_t st, const float* go, const int* amax, int B, int n, int C, float* gx);

/* VNLinearLeakyReLU on a map whose VN vectors are all parallel to ONE unit direction u per point (the layer behind VN_PointNet's
 * single-channel first layer, models/pcn.py:165-173 + models/vn_layers.py:46-74), on per-POINT operands:
 *   p[(n,t)][c] = u[n][t] v[n][c] + bias[g(n)*3+t][c],   d[(n,t)][c] = u[n][t] v[n][C+c] + bias[g(n)*3+t][C+c],   g(n) = n / points_per_group
 * v [P][ldv] = h [W_f ; W_d]^T is ONE row per point (vnp_gemm on points: a third of the rows layout's FLOPs); p, d are rebuilt in registers.
 *   mode 0: partial [nchunk][C][2] = (mean, M2) of |p|+eps per chunk of chunk_points points   (then vnp_vn_bn_finalize_chunked)
 *   mode 1: out [P*3][ldo] = leaky(bn(p), d)                                                   (rows layout)
 *   mode 2: partial [nchunk][C][2] = (sum g_nhat, sum g_nhat*xhat) from the upstream gradient go [P*3][ldgo]
 *   mode 3: gv [P][ldgv]: gv[n][c] = sum_t u_t dL/dp_t, gv[n][C+c] = sum_t u_t dL/dd_t; colsum (nullable) [nchunk][3][2C] = per-chunk
 *           sums of (dL/dp | dL/dd) per component (the bias gradient); S [C][2] = reduced mode-2 partials (NULL: eval-mode BatchNorm)
 * bias (nullable) [G*3][ldb]; with a bias chunk_points must divide points_per_group.  4 <= chunk_points <= 1024. */
int vnp_vn_dir_layer(hipStream_t st, int mode, const float* v, int ldv, const float* u, const float* bias, int ldb,
                     long long P, int C, int points_per_group, int chunk_points,
                     const float* mean, const float* rstd, const float* gamma, const float* beta, float negative_slope,
                     float* out, int ldo, const float* go, int ldgo, float* partial, const float* S, long long count,
                     float* gv, int ldgv, float* colsum);
/* the same with scalar_io != 0: the layer's OUTPUT is parallel to u as well (bias == NULL: the single-channel first layer of VN_PointNet,
 * models/pcn.py:165-167), so out [P][ldo] = sum_t u_t out_t and go [P][ldgo] (dL/dout_t = u_t go) are one scalar per (point, channel) */
int vnp_vn_dir_layer_ex(hipStream_t st, int mode, const float* v, int ldv, const float* u, const float* bias, int ldb,
                        long long P, int C, int points_per_group, int chunk_points,
                        const float* mean, const float* rstd, const float* gamma, const float* beta, float negative_slope,
                        float* out, int ldo, const float* go, int ldgo, float* partial, const float* S, long long count,
                        float* gv, int ldgv, float* colsum, int scalar_io);
/* u[n] = xyz[n] / |xyz[n]| (0 for the origin), [P][3]; _r: also r[n] = |xyz[n]| (r may be NULL) */
int vnp_unit_dirs(hipStream_t st, const float* xyz, long long P, float* u);
int vnp_unit_dirs_r(hipStream_t st, const float* xyz, long long P, float* u, float* r);
/* VNLinear -> VNMaxPool on a map u (x) h (maxpool1 behind the single-channel first layer, models/pcn.py:168-169): y [P][ldy] holds
 * y = h W^T in columns [0, C) and y' = h (Wd W)^T in [C, 2C); the score of a point is y y'.  idx [G][C] = first arg-max inside each
 * group of gs consecutive points, out [G*3][ldo] = u[n*][t] y[n*][c].  _bwd: the gradient of y has one entry per (group, channel);
 * gW (nullable) [C][ldgw] = dL/dW, gh (nullable) [G*gs][ldgh] = dL/dh (every row written); order / pts [G][C]: the channels of a
 * group sorted by their picked point, and that point (stable sort of idx, as for vnp_pool_lin_bwd). */
int vnp_dir_pool_fwd(hipStream_t st, const float* y, int ldy, const float* u, long long G, int gs, int C, int* idx, float* out, int ldo);
int vnp_dir_pool_bwd(hipStream_t st, const float* gout, int ldgo, const float* u, const int* idx, const int* order, const int* pts,
                     const float* h, int ldh, const float* W, int ldw, long long G, int gs, int C, int Ci,
                     float* gW, int ldgw, float* gh, int ldgh);
/* expand = 0: h[n][c] = sum_t u[n][t] rows[(n,t)][c];  expand = 1: rows[(n,t)][c] = u[n][t] h[n][c];  expand = 2: rows += u h */
int vnp_dir_contract(hipStream_t st, float* rows, int ldr, const float* u, float* h, int ldh, long long P, int C, int expand);

/* ---- small reductions -------------------------------------------------------------------------------------- */
/* out[r] = sum_c X[r][c]*w[c] (+ add[((r/3)/add_group)*3 + r%3]) : final VNLinear(C->1) + point_feat (pcn.py:387) */
int vnp_rowdot(hipStream_t st, const float* X, int ldx, const float* w, long long R, int C,
               const float* add, int add_group, float* out);
/* out[pc*3+t] (+)= sum_{s<S} in[(pc*S+s)*3+t] */
int vnp_group_sum3(hipStream_t st, const float* in, long long Pc, int S, float* out, int accumulate);
/* partial [nb][ceil(R/chunk)][C] = chunked column sums of X (optionally weighted per row by w[b*R + r]) */
int vnp_colsum(hipStream_t st, const float* X, long long batch_stride, long long row_stride, int nb, long long R, int C,
               const float* w, float* partial);

/* out[r][c] = u[r]*w[c] */
int vnp_outer(hipStream_t st, const float* u, const float* w, long long R, int C, float* out, int ldo);

/* ---- layout conversion at the module boundary ---------------------------------------------------------------- */
int vnp_rows_to_ref(hipStream_t st, const float* rows, int ld, int B, int C, int N, float* ref);
int vnp_ref_to_rows(hipStream_t st, const float* ref, int B, int C, int N, float* rows, int ld);

/* ---- optimizer: fused Adam over flat buffers (train.py:70, torch.optim.Adam semantics) ----------------------- */
int vnp_adam_step(hipStream_t st, float* p, const float* g, float* m, float* v, long long n,
                  float lr, float beta1, float beta2, float eps, int step, float weight_decay);
/* Gather of per-parameter gradient tensors into the flat gradient buffer (optim.FlatBuffers.gather): dst[dst_off[i] .. + numel[i]) =
 * src[i][0 .. numel[i]) for i < count, zeros where src[i] == NULL.  src / dst_off / numel are HOST arrays (the table travels in kernel
 * arguments, 48 entries per launch); sources and slices need no particular alignment (16-byte aligned pairs move as float4). */
int vnp_pack_tensors(hipStream_t st, int count, const void* const* src, const long long* dst_off, const long long* numel, float* dst);

#ifdef __cplusplus
}
#endif
#endif /* VNPCC_H */
