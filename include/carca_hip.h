/* carca_hip.h -- C ABI of libcarca_hip.so: the MI355X (gfx950) CARCA hot path.
 *
 * The reference (r-papso/carca-replication) is pure PyTorch and has no FFI of its own; its
 * plug-in boundary is the nn.Module ABCs of src/abstract.py:8-50.  This library sits UNDER
 * those modules: each entry point replaces the ATen op sequence of one reference method, named
 * below with file:line.  The Python host layer (carca_replication_amd/modules.py) keeps the
 * reference's class names, constructors, state_dict keys and forward() semantics and binds
 * these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 (ids: int32) unless it says "host";
 *  - tensors are row-major; `ld*` is a row stride in elements; activation strides must be
 *    multiples of 4 elements and bases 16-byte aligned (torch allocations are);
 *  - `stream` is a hipStream_t passed as void*; launches are asynchronous on it and the host never
 *    waits for the device; callers own every tensor, nothing of theirs is retained after the call
 *    returns (graph-capturable).  What a launch needs beyond its arguments (partial tiles of a split
 *    product, row tables) is the library's own: one lazily grown scratch buffer per stream, reused
 *    in stream order, or memory owned by the capture (carca_capture_scope below);
 *  - return 0 on success, a negative CARCA_ERR_* for a rejected call (message from
 *    carca_last_error()), a positive hipError_t for a failed launch.  Never aborts.
 */
#ifndef CARCA_HIP_H
#define CARCA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CARCA_ABI_VERSION 2
#define CARCA_OK 0
#define CARCA_ERR_UNSUPPORTED (-1) /* shape outside what the kernels were built for */
#define CARCA_ERR_BADARG (-2)      /* null pointer, misaligned stride, ... */

#define CARCA_MAX_SEGS 4   /* profile + up to 3 target groups per embed call */
#define CARCA_MAX_GROUPS 3 /* target groups per scoring call (train: pos, neg) */
#define CARCA_MAX_L 64     /* profile slots one workgroup keeps in LDS */
#define CARCA_EMBED_GATHER 1
#define CARCA_EMBED_FEAT 2
#define CARCA_EMBED_JOINT 4
#define CARCA_EMBED_ALL 7

int carca_abi_version(void);
/* Kernel-variant knobs for tuning runs and A/B tests (tools/, tests/); 0 = the shipped choice everywhere.
 *   key 0  row / weight-gradient GEMM: 1 force 128x96 tiles, 2 force the one-block-per-CU kernels, 3 = 2 + in-kernel
 *          stamps, 4 no buffer loads, 5 tiled weight gradient only, 6 never group weight gradients / row GEMMs,
 *          7 force the one-block-per-CU row GEMM with 384 x 128 tiles, 8 never let the item-row gather ride in the
 *          feature GEMM's launch, 15 the one-tile-per-workgroup feature GEMM (no hand-over of partial tiles), 19 the gather
 *          keeps its own launch beside the stream-K kernels, 21 split-precision kernel without LDS-DMA, 22 weight gradient
 *          over every row, 23 feature GEMM over every row (gemm_rows_sk_kernel instead of gemm_rows_skc_kernel, which
 *          leaves rows with id 0 out); any non-zero value also moves the narrow row products off their default kernel
 *   key 1  attention kernels: 1 one workgroup per user, 2 always two, 3 one 8-wave workgroup per user (the variant
 *          for batches of more users than CUs, two workgroups resident per CU)
 *   key 2  weight gradient: row-split slot target      key 4  weight gradient: minimum 32-row chunks per split
 *   key 3  weight gradient: plain stores instead of atomics (timing diagnostic, wrong results)
 *   key 5  grouped weight gradient: row-split slot target per product
 *   key 6  1: the round-1 paths (materialised V in the scoring kernel, per-op SelfAttentionBlock backward)
 *   key 7  scoring kernel layout (1 one workgroup per user, 2 fold kernel, 3 persistent stream kernel)
 *   key 8  DETERMINISTIC MODE (1 = on): no fp32 atomics anywhere in the backward pass -- every accumulation into the
 *          pass's gradient buffer (weight gradients, LayerNorm gamma / beta, bias sums, the embedding scatter-add) goes
 *          into a 64-bit fixed-point shadow buffer (carca_det_begin / carca_det_flush below), whose integer sums do not
 *          depend on the order in which workgroups arrive; HR / NDCG sums are added in a fixed order by one block.
 *          Same inputs => same bits, run to run.  Reference: none (torch's CUDA backward has the same nondeterminism:
 *          index_add / atomicAdd in embedding_dense_backward); it exists so that trajectory tests can be tight.
 *   key 9  in-kernel phase stamps of the row-chain kernels (1 FFN side, 2 input side; carca_set_debug_buffer)
 *   key 10 CU budget of the one-workgroup-per-CU weight-gradient kernel (0 = every CU): a caller that runs two halves
 *          of a backward pass on two streams gives each launch half the chip (autograd.py: the target rows' embedding
 *          backward beside the profile rows' encoder backward)
 *   key 11 feature GEMM with the stream-K hand-over: K steps the cheap workgroup takes over per tile (0 = cost model)
 *   key 12 ... log2 of the bound on a taker's wait for its partial tile, in ~0.4 us sleeps (0 = 23, about 5 s); on expiry
 *          the launch's output is wrong, the library's error word is set and carca_poll_errors / the next such launch fail
 *   key 13 ... TEST ONLY: 1 + index of the one partial tile whose giver withholds its flag (exercises key 12's expiry)
 *   key 14 persistent scoring kernel: 1 = ticket dealing of a step's first jobs (A/B)
 *   key 15 timing diagnostics of the 80 x 96 row GEMM, the scoring kernels, the split-precision kernels and the prologue of
 *          gemm_rows_skc_kernel (bit masks; WRONG results)
 *   key 16 PRECISION OF THE FEATURE GEMM (AllEmbedding.feats_embed, carca.py:86), opt-in: 0 = exact-fp32 MFMA (default,
 *          the path every parity figure is quoted on); 1 = operands split into three bf16 parts, six bf16-MFMA products,
 *          fp32 accumulation (fp32-class accuracy on the 16x faster pipe); 2 = two fp16 parts (the second scaled by
 *          2^11), three products, two fp32 accumulators -- |operands| < 65504 required.  Only where the one-workgroup-per-CU
 *          kernel would run; anything else keeps the fp32 kernels.  + 16: wherever the kernel's own conditions hold
 *          (K0 % 4 == 0, K1 <= 8, plain epilogue), whatever the grid -- for parity tests at fixture sizes. */
int carca_set_tuning(int key, int value);
/* Deterministic mode, per backward pass: register the pass's flat fp32 gradient buffer `flat` (n floats) and its shadow
 * (n uint64, ZERO on entry); kernels launched on `stream` afterwards accumulate gradients that land inside `flat` into
 * the shadow.  carca_det_flush(lo, hi) adds shadow[lo:hi] / 2^36 into flat[lo:hi] and clears it -- before anything
 * READS an accumulated range and at the end of the pass.  carca_det_begin(NULL, 0, NULL) ends the pass. */
int carca_det_begin(float* flat, long long n, unsigned long long* shadow, void* stream);
int carca_det_flush(float* flat, unsigned long long* shadow, long long lo, long long hi, void* stream);
/* Diagnostic runs only: device buffer (>= 16 x #workgroups uint64) that the attention kernels fill with
 * s_memtime stamps at their phase boundaries; NULL (default) disables stamping. */
int carca_set_debug_buffer(void* device_ptr);
const char* carca_last_error(void); /* host string, thread-local, valid until the next call */
/* Failures of a kernel that no launch status can carry (today: key 12's expiry).  CARCA_OK, or a negative code with the
 * message set; the condition is cleared by the call that reports it.  Costs one read of host memory: call it wherever the
 * host has synchronised anyway (end of an evaluation epoch, after a graph replay's loss was read). */
int carca_poll_errors(void);
/* Memory the library allocates for kernels that are CAPTURED into a hipGraph (partial-tile buffers, row tables, descriptor
 * copies -- what an eager launch takes from a per-stream scratch buffer) belongs to the capture in progress:
 * carca_capture_scope returns that capture's id while `stream` is capturing (0 otherwise), carca_capture_bytes what has
 * been allocated under it, and carca_capture_release frees it -- call it once the graph built from the capture is gone. */
int carca_capture_scope(void* stream, unsigned long long* id_out);
long long carca_capture_bytes(unsigned long long id);
int carca_capture_release(unsigned long long id);
/* Outside a capture the same memory is one buffer per (device, stream, purpose), kept for the life of the process and keyed
 * by the raw stream handle: a caller that DESTROYS a stream it launched on hands the stream's buffers back with this call
 * (after synchronising the stream, on the device the launches ran on; refused while the stream is being captured). */
int carca_release_stream_scratch(void* stream);

/* ------------------------------------------------------------------------------------------
 * Padded geometry shared by every entry point.  For model width d and H heads (dh = d/H):
 *   DPI = 64 / 96 / 128 (smallest >= d) padded input-feature count (k of the projections)
 *   DHP = round_up(dh, 16)           padded per-head width
 *   DPO = H * DHP                    head-padded output-feature count
 * Activations travel between kernels as [rows, DPI] with zeroed pad columns.
 * ---------------------------------------------------------------------------------------- */
int carca_padded_dims(int d, int H, int* dpi, int* dhp, int* dpo);

/* ---- weight packing -----------------------------------------------------------------------
 * Copies parameter matrices (state_dict layout) into the zero-padded, 16-byte-aligned,
 * optionally head-permuted layout the attention kernels read MFMA fragments from.
 * dst[rp][cp] = src[r][c] where rp (cp) is r (c) itself, or its head-padded position when
 * row_dh (col_dh) > 0; every dst element is written (pads = 0).  descs is a HOST array. */
typedef struct CarcaPackDesc {
  const float* src; /* [rows, cols], row stride src_ld */
  float* dst;       /* [dst_rows, dst_cols] dense */
  int32_t rows, cols, src_ld;
  int32_t dst_rows, dst_cols;
  int32_t row_dh, row_dhp; /* 0,0 = plain rows */
  int32_t col_dh, col_dhp; /* 0,0 = plain cols */
  int32_t transposed;      /* 1: logical element (r, c) lives at src[c * src_ld + r] (rows/cols are logical) */
  int32_t frag16;          /* 1: dst is written in MFMA-fragment order instead of row-major (dst_rows, dst_cols multiples
                            * of 16): packed element (rp, cp) lives at
                            *   (((rp/16) * (dst_cols/16) + cp/16) * 64 + ((cp%16)/4) * 16 + rp%16) * 4 + cp%4,
                            * i.e. the 64 lanes x 4 floats one wave loads for one 16x16 tile are 1 KB contiguous */
  /* fold_vec != NULL: the logical source is not src itself but the per-head contraction of its ROWS with a vector,
   *   S'[h][c] = sum_{i < rows/fold_H} fold_vec[h*(rows/fold_H) + i] * src[h*(rows/fold_H) + i][c],  h < fold_H,
   * a [fold_H, cols] matrix that is then placed like any other (row_dh must be 0).  This is how decoder.ffn is folded
   * into the cross-attention's value projection once per weight version (CarcaCaWeights.wu / cu). */
  const float* fold_vec;
  int32_t fold_H;
} CarcaPackDesc;
int carca_pack_weights(const CarcaPackDesc* descs, int n, void* stream);

/* ---- building block: row GEMM  C[m][n] = sum_k A[m][k] * Bt[n][k]  (+ epilogue) ------------------
 * The tiled fp32-MFMA kernel behind AllEmbedding's two Linear layers (carca.py:86,89) and every
 * dense input-gradient product of the backward pass.  Rows come from up to CARCA_MAX_SEGS segments
 * (one launch for profile + target groups); k runs over up to two column sources (a0 | a1), each
 * with its own Bt, so torch.cat((a, c), -1) @ W^T never materialises the concatenation.
 * Epilogue, in this order, each part optional:
 *   v = alpha * acc + bias[n] + pos[(row % T)][n] + add[row][n] + add_table[ids[row]][n] + rowscale[row] * colvec[n]
 *   v *= gate_scale * (gate[row][n] > 0 ? 1 : gate_slope) (LeakyReLU' (x dropout1'), from the saved activation)
 *   v  = ids[row] != 0 ? v : 0                          (when mask_rows)
 *   C[row][n] = v for n < N;  C[row][n] = 0 for N <= n < ncols_out                              */
typedef struct CarcaGemmSeg {
  const float* a0;       /* [rows, lda0] */
  const float* a1;       /* [rows, lda1] or NULL when K1 == 0 */
  float* c;              /* [rows, ldc] */
  const int32_t* ids;    /* [rows] or NULL (needed by mask_rows / add_pos) */
  const float* add;      /* [rows, ld_add] or NULL */
  const float* gate;     /* [rows, ld_gate] or NULL */
  const float* rowscale; /* [rows] or NULL */
  int32_t rows, T, add_pos;
  /* Rows of a0 / a1 may belong to a [B, T, K] VIEW whose users are a0_bstride / a1_bstride elements apart
   * (e.g. o_a[:, :L] of train.py:86-88): row r then starts at (r / T) * bstride + (r % T) * lda.  0 = dense. */
  int64_t a0_bstride, a1_bstride;
  /* >= 1: a0 is a TABLE [n_items, lda0] and row r reads a0[ids[r]] (attribute rows gathered by item id inside the
   * GEMM's operand load -- no dense [B, T, n_attrs] tensor exists; data.py:119-132 always sets p_a = attrs[p_x]).
   * A value > 1 also states the table's row count, which lets the launcher pick the 32-bit-offset load path. */
  int32_t a0_gather;
} CarcaGemmSeg;
typedef struct CarcaGemmDesc {
  CarcaGemmSeg seg[CARCA_MAX_SEGS];
  int32_t nseg;
  int32_t lda0, lda1, K0, K1;
  const float* bt0; /* [N, ldb0] */
  const float* bt1; /* [N, ldb1] or NULL */
  int32_t ldb0, ldb1;
  int32_t N, ldc, ncols_out;
  const float* bias;   /* [N] or NULL */
  const float* pos;    /* [T, N] or NULL */
  const float* colvec; /* [N] or NULL */
  int32_t ld_add, ld_gate;
  float gate_slope;
  int32_t mask_rows;
  float alpha;             /* v = alpha * acc + bias + ...; 0 means 1 */
  float gate_scale;        /* extra factor on the gate (1/(1-p) when the saved activation went through dropout); 0 = 1 */
  int32_t gate_zero_drops; /* 1: an exactly-zero saved activation was DROPPED -> gradient 0 (else LeakyReLU'(0) = slope) */
  /* An addend GATHERED by id: v += add_table[ids[row]][n] (every segment needs ids).  How the joint embedding takes the
   * item rows without a gather launch and without their K columns: e = q W_jq^T + (sqrt(d) E W_jz^T)[id] + b_j
   * (carca.py:87-89), the bracket prepared once per weight version (CarcaForwardDesc.z_table). */
  const float* add_table;  /* [n_rows_of_the_table, ld_add_table] or NULL */
  int32_t ld_add_table;
} CarcaGemmDesc;
int carca_gemm_rows(const CarcaGemmDesc* desc /*host*/, void* stream);
/* Kernels behind it and what they assume of the GPU (the launcher picks by shape; carca_gemm_rows_log names the choice):
 *  - gemm_rows_sk_kernel / gemm_rows_skc_kernel (K0 >= 2048 with a grid of about one 384 x 96 tile per CU: the feature
 *    product at C2 / C3 / C4) are PERSISTENT grids of one workgroup per CU in which a workgroup may WAIT for a partial tile
 *    that another workgroup of the same launch writes (stream-K hand-over).  Every workgroup of the launch must become
 *    resident: give such a launch the whole device -- ONE stream of this library per device at a time, no CU mask smaller
 *    than the grid, no long-running kernel of another stream holding CUs.  The wait is bounded (tuning key 12, ~5 s): a
 *    taker that gives up writes the library's error word, and carca_poll_errors / the next such launch fail loudly instead
 *    of the GPU hanging; the results of that launch are wrong.  The captured train step's second stream runs its kernels
 *    under a CU budget (tuning key 10) and never one of these two beside another.
 *  - gemm_rows_cus_kernel (short K, >= 1.75 tiles per CU: the feature product at C5) and gemm_rows_n96s_kernel (narrow
 *    output over >= 2.5 blocks of 160 rows per CU: the joint embedding at C5) are persistent too, but no workgroup ever
 *    waits for another: they are correct under any scheduling, beside any other stream.
 *  - every other kernel is an ordinary grid of independent tiles. */
/* Opt-in split-precision path of the product above (tuning key 16; csrc/gemm_split.hip): the weight matrix as packed
 * 16-bit planes, prepared once per weight version.  It takes products with K0 % 4 == 0 (k-source 0 in 32-wide K steps;
 * k-source 1, at most 8 columns, is added as exact fp32 multiply-adds) and the plain epilogue, on 384 x 96 tiles.
 * carca_split_bytes gives the size of the packed copy of the [N, K0] block of k-source 0 (mode 1 = bf16 x 3, 2 = fp16 x 2),
 * carca_split_pack writes it (w = bt0, row stride ldw), and carca_split_bind tells this THREAD's following launches that
 * `planes` is the packed copy of the matrix at `w`: a launch whose bt0, mode and shape match uses it, any other launch
 * under key 16 splits its weights itself, into scratch, every time (correct, ~8 us at C2).  carca_split_bind(NULL, NULL,
 * 0, 0, 0) unbinds.  The caller keeps `planes` alive and re-packs when the weights change. */
long long carca_split_bytes(int N, int K0, int mode);
int carca_split_pack(const float* w, int ldw, int K0, int N, int mode, void* out, void* stream);
int carca_split_bind(const float* w, const void* planes, int mode, int N, int K0);
long long carca_split_launch_count(void); /* launches of the split-precision kernel by this process so far */
/* n INDEPENDENT products (host array): the narrow ones (d-wide input gradients of the backward pass) share launches,
 * so that two ~150-block products cost one launch instead of two back to back; same results as n single calls. */
int carca_gemm_rows_group(const CarcaGemmDesc* descs /*host*/, int n, void* stream);

/* ---- building block: weight-gradient GEMM  dW[n][k] += sum_r dY[r][n] * X[r][k] -------------------
 * The contraction runs over ROWS (users x slots), split across workgroups; partial tiles are
 * combined with fp32 atomic adds into dW (caller zeroes dW / db first; summation order, hence the
 * last bits, varies run to run).  db[n] += sum_r dY[r][n] when db != NULL.  Rows whose ids == 0
 * are skipped when mask_rows (the e * mask of carca.py:94 seen from the backward side). */
typedef struct CarcaWgradSeg {
  const float* dy;    /* [rows, ld_dy] */
  const float* x;     /* [rows, ld_x]  -> dW[:, 0:K] */
  const float* x1;    /* [rows, ld_x1] -> dW[:, K:K+K1], or NULL when K1 == 0 */
  const int32_t* ids; /* [rows] or NULL */
  int32_t rows;
  int32_t T;                     /* rows per user, used with the strides below */
  int64_t x_bstride, x1_bstride; /* users of a [B, T, K] view are this many elements apart; 0 = dense */
  int32_t x_gather;              /* >= 1: x is a table [n_items, ld_x], row r reads x[ids[r]]; > 1 also states n_items */
} CarcaWgradSeg;
typedef struct CarcaWgradDesc {
  CarcaWgradSeg seg[CARCA_MAX_SEGS];
  int32_t nseg;
  int32_t ld_dy, ld_x, ld_x1;
  int32_t N, K, K1; /* dW is [N, K + K1] */
  float* dw;
  int32_t ldw;
  float* db; /* [N] or NULL */
  int32_t mask_rows;
} CarcaWgradDesc;
int carca_gemm_wgrad(const CarcaWgradDesc* desc /*host*/, void* stream);
/* n independent products in ONE launch (the d x d weight gradients of a backward pass are ~20 us latency-bound
 * launches; side by side they fill the chip).  Same semantics as n calls of carca_gemm_wgrad in any order; products
 * with more than 96 x 1024 outputs are passed on to carca_gemm_wgrad one by one. */
int carca_gemm_wgrad_group(const CarcaWgradDesc* descs /*host, [n]*/, int n, void* stream);

/* ---- a1 + a2 + a9: AllEmbedding.forward over several row segments ---------------------------
 * Replaces get_mask (utils.py:6-7) + AllEmbedding.forward (carca.py:85-95) + the additive
 * encodings (carca.py:25-31, 54-60) for the profile and every target group in ONE call:
 *   q = [attrs ; ctx] W_f^T + b_f ; z = E[ids] * sqrt(d) ; e = [z ; q] W_j^T + b_j (+ pos[t]) ;
 *   e *= (ids != 0).
 * zq is workspace AND the tensor the backward needs: [sum(rows), d + g], rows of segment s
 * start at sum(rows of earlier segments).  The q columns of rows with id 0 are written as ZEROS, not computed: their e is
 * masked (carca.py:92-94) and their gradient is zero, so the feature product leaves those rows out where that pays
 * (gemm_rows_skc_kernel: 16 % of an evaluation batch's rows at BASELINE's profile lengths, 47 % of a training batch's).
 * e_out rows have stride ld_e >= d, columns d..ld_e-1
 * are written as zeros.  `stages` selects which of the three launches to issue (bit 0 gather,
 * bit 1 feature GEMM, bit 2 joint GEMM; 7 = all) so a profiler can bracket one of them with events. */
typedef struct CarcaRowSeg {
  const int32_t* ids; /* [rows] item ids, 0 = pad */
  const float* attrs; /* [rows, n_attrs] */
  const float* ctx;   /* [rows, n_ctx] */
  float* e_out;       /* [rows, ld_e] */
  int32_t rows;       /* B * T */
  int32_t T;          /* slots per user (position index = row % T) */
  int32_t add_pos;    /* 1: add pos[row % T] (profile side, carca.py:91-92) */
  int64_t attrs_bstride, ctx_bstride; /* elements between users when attrs/ctx are [B, T, .] views; 0 = dense */
  const float* attrs_table; /* optional [n_items, n_attrs]: when set, `attrs` is ignored and row r uses attrs_table[ids[r]] */
  int32_t attrs_table_rows; /* n_items of attrs_table when known, else 0 (only a speed hint: see CarcaGemmSeg.a0_gather) */
} CarcaRowSeg;
int carca_embed_fwd(const CarcaRowSeg* segs /*host*/, int nseg, int n_attrs, int n_ctx, int d, int g,
                    const float* items_w /*[n_items,d]*/, const float* feats_w /*[g,n_attrs+n_ctx]*/,
                    const float* feats_b /*[g]*/, const float* joint_w /*[d,d+g]*/, const float* joint_b /*[d]*/,
                    const float* pos /*[T,d] or NULL*/, float* zq, int ld_e, int stages, void* stream);

/* ---- a3 + a4: SelfAttentionBlock.forward -----------------------------------------------------
 * Replaces SelfAttentionBlock.forward (carca.py:297-318) incl. MultiHeadAttention.forward
 * (carca.py:228-265) with causal=0, in eval mode / dropout p = 0.  One workgroup per user.
 * Pointers in CarcaSaWeights are PACKED (carca_pack_weights): projections [DPO, DPI] with
 * head-padded rows, biases [DPO] head-padded, ffn matrices [DPI, DPI], LayerNorm vectors [DPI].
 * The five matrices are in FRAGMENT ORDER (CarcaPackDesc.frag16 = 1): a wave's operand load for one 16x16 tile is
 * then eight full cache lines instead of sixteen half-used ones (row-major cost 15 k of the 21 k cycles of the
 * K/V projection phase at C2). */
typedef struct CarcaSaWeights {
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b; /* [DPI] */
  const float *wq, *wk, *wv;                  /* [DPO, DPI], fragment order */
  const float *bq, *bk, *bv;                  /* [DPO] */
  const float *w1, *w2;                       /* [DPI, DPI], fragment order */
  const float *b1, *b2;                       /* [DPI] */
} CarcaSaWeights;
/* Training-mode dropout (nn.Dropout sites carca.py:258,309,312,416): element e of a site is kept iff
 * hash(seed, site, e) >= p * 2^24 (counter-based: no state, any launch order), kept values are scaled by
 * 1/(1-p).  torch's Philox stream cannot be reproduced inside a kernel, so parity is defined at p = 0 and
 * through the masks: every site also writes its keep-mask (uint8, 1 = kept) so that tests can replay the
 * reference arithmetic with the SAME masks.  p = 0 (or a NULL pointer) disables everything. */
typedef struct CarcaDropout {
  float p;
  uint64_t seed;
  uint32_t site; /* first site id of this call; a kernel with several sites uses site, site+1, ... */
  const uint64_t* seed_offset; /* device, or NULL: *seed_offset is added to `seed` when the kernel starts -- a step
                                  captured into a hipGraph repeats its launch arguments, the graph bumps this counter */
} CarcaDropout;

/* Tensors the backward pass needs (all optional; pass save = NULL in eval): */
typedef struct CarcaSaSave {
  float* qn;             /* [B*L, DPI] LayerNorm1(x) */
  float *qh, *kh, *vh;   /* [B*L, DPO] projections, head-padded columns */
  float* r;              /* [B*L, DPI] LayerNorm2's input (attention + residual) */
  float* s2;             /* [B*L, DPI] LayerNorm2's output */
  float* h1;             /* [B*L, DPI] dropout1(LeakyReLU(ffn_1(s2))) */
  uint8_t* m_attn;       /* [B, H, L, L] keep-mask of the attention-weight dropout (site+0), or NULL */
  uint8_t* m_ffn1;       /* [B*L, DPI] keep-mask of dropout1 (site+1), or NULL */
  uint8_t* m_ffn2;       /* [B*L, DPI] keep-mask of dropout2 (site+2), or NULL */
} CarcaSaSave;
int carca_sa_block_fwd(const float* x /*[B*L, ldx]*/, int ldx, const int32_t* ids /*[B*L]*/, float* y /*[B*L, ldy]*/,
                       int ldy, int B, int L, int d, int H, const CarcaSaWeights* w /*host struct*/, int residual,
                       const CarcaSaSave* save /*host struct or NULL*/, const CarcaDropout* drop /*or NULL*/,
                       void* stream);

/* The same block in eval mode (nothing saved, no dropout) with one more promise from the caller:
 *   pads_uniform != 0: within a user, all LEADING pad rows (ids == 0 before the first real slot) of x are equal.  True for
 *   the masked embedding (rows of zeros, carca.py:94) and, since a pad row's output depends on its own input only,
 *   for every block output downstream of it; carca_forward passes 1.  The kernel then computes one of those rows and
 *   writes it to every leading pad slot, and projects / attends / feeds forward only ceil((real slots + 1) / 16) row
 *   tiles.  0 = no assumption (what carca_sa_block_fwd passes for save == NULL, drop == NULL). */
int carca_sa_block_eval(const float* x, int ldx, const int32_t* ids, float* y, int ldy, int B, int L, int d, int H,
                        const CarcaSaWeights* w /*host struct*/, int residual, int pads_uniform, void* stream);

/* ---- a5 + a6: final LayerNorm + CrossAttentionBlock.forward, grouped --------------------------
 * Replaces CARCA.forward's final norm (carca.py:421) and, for every target group,
 * CrossAttentionBlock.forward (carca.py:338-349): K/V projections of the normed profile are
 * computed once per user, then each group's targets are scored:
 *   y = sigmoid(ffn(MHA(o, p, p) + o)), tril(-1) masking iff training (carca.py:339).
 * One workgroup per user.  y_out[g] is [B, N_g] dense. */
typedef struct CarcaCaWeights {
  const float *ln_w, *ln_b; /* final norm [DPI]; both NULL = p_raw is already normed (standalone decoder) */
  const float *wq, *wk, *wv; /* [DPO, DPI], fragment order (CarcaPackDesc.frag16) */
  const float *bq, *bk, *bv; /* [DPO] */
  const float* ffn_w_pad;    /* decoder.ffn.weight in head-padded order [DPO] */
  const float* ffn_w;        /* decoder.ffn.weight plain, zero-padded [DPI] */
  const float* ffn_b;        /* [1] */
  /* decoder.ffn folded into the value projection (the block's output is the scalar w . (PV + o), carca.py:340-345, so
   * w_h . (P_h V_h) = P_h u_h with u_h[key] = p[key] . wu[h] + cu[h]): wu[h][c] = sum_i w[h*dh+i] W_V[h*dh+i][c]
   * ([16, DPI], rows >= H zero, fragment order), cu[h] = sum_i w[h*dh+i] b_V[h*dh+i] ([16]).  Built by
   * carca_pack_weights (CarcaPackDesc.fold_vec).  Used by the inference kernel (save == NULL); may be NULL, which
   * selects the kernel that materialises V. */
  const float *wu, *cu;
} CarcaCaWeights;
typedef struct CarcaTargetGroup {
  const float* o;     /* embedded targets [B*N, ldo] */
  const int32_t* ids; /* [B*N] */
  float* y;           /* [B, N], row stride ldy */
  int32_t N;
  int32_t ldy;        /* elements between users in y; 0 = N (dense).  > N: the groups' scores are column blocks of ONE
                       * [B, sum N] tensor (what CARCA.forward's torch.cat would build, carca.py:431) */
} CarcaTargetGroup;
typedef struct CarcaCaSave {
  float *kh, *vh;              /* [B*L, DPO] */
  float* qh[CARCA_MAX_GROUPS]; /* [B*N_g, DPO] per group */
  uint8_t* m_attn[CARCA_MAX_GROUPS]; /* [B, H, N_g, L] keep-masks of the attention-weight dropout (site+g), or NULL */
} CarcaCaSave;
int carca_cross_score_fwd(const float* p_raw /*[B*L, ldp] encoder output BEFORE the final norm*/, int ldp,
                          const int32_t* p_ids /*[B*L]*/, float* p_normed /*[B*L, ldp] or NULL*/,
                          const CarcaTargetGroup* groups /*host*/, int ngroups, int ldo, int B, int L, int d, int H,
                          const CarcaCaWeights* w /*host struct*/, int residual, int training,
                          const CarcaCaSave* save /*host struct or NULL*/, const CarcaDropout* drop /*or NULL*/,
                          void* stream);

/* ---- backward kernels (the autograd of the lines cited at each forward entry point) -------------------
 * LayerNorm backward over `rows` rows of width d: dx = dLN(dy; x, gamma) (+ addend); dgamma/dbeta are
 * accumulated with atomics (caller zeroes).  Columns d..ncols_out-1 of dx are written as zeros. */
int carca_layernorm_bwd(const float* dy, int ld_dy, const float* x, int ld_x, const float* gamma, int rows, int d,
                        const float* addend /*or NULL*/, int ld_add, float* dx, int ld_dx, int ncols_out,
                        float* dgamma /*or NULL*/, float* dbeta /*or NULL*/, void* stream);
/* d_items[ids[r]][:] += scale * dz[r][:] for ids[r] != 0 (nn.Embedding(padding_idx=0), carca.py:73,87-88) */
int carca_embed_scatter(const float* dz, int ld_dz, const int32_t* ids, int rows, int d, float scale,
                        float* d_items /*[n_items, d]*/, void* stream);
/* out[(row % T)][c] += sum_rows rowscale[row] * (ids[row] != 0) * x[row][c]; T = 1: plain column sum.
 * rowscale / ids may be NULL.  cols <= 256.  Caller zeroes out. */
int carca_colsum(const float* x, int ld_x, int rows, int cols, const float* rowscale, const int32_t* ids, int T,
                 float* out /*[T, cols]*/, void* stream);
/* Attention core of SelfAttentionBlock (carca.py:246-260, causal = 0): given d(attention output)
 * [B*L, ld_da] in plain feature order and the saved projections, recompute P and return dQ, dK, dV
 * (head-padded, [B*L, DPO]). */
int carca_sa_attn_bwd(const float* qh, const float* kh, const float* vh, const float* d_attn, int ld_da,
                      const int32_t* ids, float* dqh, float* dkh, float* dvh, int B, int L, int d, int H,
                      const uint8_t* m_attn /*[B,H,L,L] or NULL*/, float drop_scale /*1/(1-p)*/, void* stream);
/* Attention core + sigmoid(ffn(.)) head of CrossAttentionBlock (carca.py:340-347) for every group:
 * dlogit = dy * y * (1 - y); d(attention output) = dlogit (x) ffn_w_pad; returns dQ per group, dK, dV
 * (summed over groups) and accumulates d ffn_w_pad[f] += sum dlogit * O[.][f] (caller zeroes it). */
typedef struct CarcaCrossBwdGroup {
  const float* qh;    /* [B*N, DPO] saved by the forward */
  const float* y;     /* [B*N] forward output */
  const float* dy;    /* [B*N] incoming gradient */
  const int32_t* ids; /* [B*N] */
  float* dqh;         /* [B*N, DPO] out */
  float* dlogit;      /* [B*N] out, or NULL */
  const uint8_t* m_attn; /* [B, H, N, L] keep-mask saved by the forward, or NULL */
  int32_t N;
  int32_t ld_y;          /* elements between users in y AND dy; 0 = N (see CarcaTargetGroup.ldy) */
} CarcaCrossBwdGroup;
int carca_cross_attn_bwd(const float* kh, const float* vh, const int32_t* p_ids, const CarcaCrossBwdGroup* groups,
                         int ngroups, const float* ffn_w_pad, float* dkh, float* dvh, float* d_ffn_w_pad, int B, int L,
                         int d, int H, int training, float drop_scale, void* stream);
/* In-place dropout of x [rows, cols] (row stride ld): x = keep ? x / (1-p) : 0; mask [rows, cols] uint8.
 * Used for CARCA.dropout on the profile embedding (carca.py:416). */
int carca_dropout_fwd(float* x, int rows, int cols, int ld, const CarcaDropout* drop, uint8_t* mask, void* stream);
/* out[r][c] = x[r][c] * mask[r][c] * scale for c < cols, 0 for cols <= c < ncols_out (dropout backward) */
int carca_mask_mul(const float* x, int ld_x, const uint8_t* mask, int ld_m, float scale, float* out, int ld_out,
                   int rows, int cols, int ncols_out, void* stream);
/* ---- f4: the reference's ablation decoders and the stand-alone final LayerNorm ---------------------------
 * Row kernels on the padded activation layout (d <= 128; pad columns of every output are written as zeros).
 * carca_layernorm_fwd replaces CARCA.norm.forward (carca.py:421) where no cross-attention kernel fuses it. */
int carca_layernorm_fwd(const float* x, int ldx, float* y, int ldy, int rows, int d, const float* w, const float* b,
                        void* stream);
/* DotProduct.forward (carca.py:361-367) / the scoring step of WeightedDotProduct.forward (carca.py:390-397):
 * y[b][t] = link(p_row . o[b][t]) with p_row = p[b][t] (slotwise = 1: train mode, needs T == L) or p[b][L-1]
 * (slotwise = 0: eval mode); link 0 = sigmoid, 1 = (s + 1) / 2.  p is [B*L, ldp], o is [B*T, ldo], y is [B*T]. */
int carca_dot_score_fwd(const float* p, int ldp, const float* o, int ldo, float* y, int B, int L, int T, int d,
                        int slotwise, int link, void* stream);
/* its backward: d_o[row] = dl * p_row (written), dp[p_row] += dl * o[row] (ACCUMULATED: caller zeroes dp before the
 * first group), dl = dy * dlink/ds */
int carca_dot_score_bwd(const float* p, int ldp, const float* o, int ldo, const float* y, const float* dy, float* dp,
                        int ld_dp, float* d_o, int ld_do, int B, int L, int T, int d, int slotwise, int link,
                        void* stream);
/* WeightedDotProduct's history weighting (carca.py:376-378,385-386): out[b][t][:] = c_t x[b][t][:] with
 * c_t = sum_{j<=t} gamma^j (the reference repeats the history along a new axis, so slots are scaled, not mixed).
 * The map is diagonal, hence its own backward. */
int carca_slot_decay_scale(const float* x, int ldx, float* out, int ldo, int B, int L, int d, float gamma, void* stream);
/* torch.nn.functional.normalize(x, dim=-1) (carca.py:388-389) and its backward */
int carca_l2norm_fwd(const float* x, int ldx, float* y, int ldy, int rows, int d, void* stream);
int carca_l2norm_bwd(const float* x, int ldx, const float* dy, int ld_dy, float* dx, int ld_dx, int rows, int d,
                     void* stream);
/* KNN.forward (knn.py:13-19), the reference's attribute-similarity baseline model: y[b][t] = p_last(b) . o(b, t) over
 * the F attribute features, p_last = the LAST profile slot's attribute row (knn.py:14); raw dot product, no link.
 *   table_rows == 0: p_a is the dense [B, L, F] profile attributes, o_a the dense [B, T, F] target attributes, rows
 *                    contiguous, user b at p_a + b*p_bstride / o_a + b*o_bstride floats (so the two train groups may
 *                    be the halves of one [B, 2L, F] tensor, train.py:86-88); p_x / o_x unused, may be NULL;
 *   table_rows  > 0: p_a is the attribute table [table_rows, F] (o_a unused); rows are p_x[b][L-1] and o_x[b][t]
 *                    (p_x [B, L], o_x [B*T], both contiguous; strides unused); ids outside the table score 0.
 * HBM-bound: B*T*F*4 bytes streamed once. */
int carca_knn_score(const float* p_a, int64_t p_bstride, const float* o_a, int64_t o_bstride, const int32_t* p_x,
                    const int32_t* o_x, int table_rows, float* y, int B, int L, int T, int F, void* stream);
/* ---- stand-alone pieces of the reference's module surface (abstract.py:31, carca.py:25-31, 54-60, 228-265) ----------
 * The hot path never runs these (encodings ride in the embedding GEMM's epilogue, attention is fused into K2 / K4);
 * they exist so that a caller who uses the modules the way the ABCs allow gets the reference's numbers instead of an
 * exception.
 * carca_add_positions: Encoding.forward(x): out[b][t][:] = x[b][t][:] + pos[t][:] for x [B*T, ldx], pos [T, d]. */
int carca_add_positions(const float* x, int ldx, const float* pos, float* out, int ldo, int B, int T, int d, void* stream);
/* carca_mha_core: the attention core of MultiHeadAttention.forward (carca.py:242-260) on PROJECTED inputs (plain feature
 * order, head h = columns [h d/H, (h+1) d/H)): q [B*Tq, ldq], k, v [B*Tk, ldk]; mask = (q_ids != 0) x (k_ids != 0),
 * lower-triangular with diagonal `causal` when has_causal; W = softmax((mask ? 0 : -2^32+1) + q k^T) / sqrt(d/H)) * mask;
 * out [B*Tq, ldo] = W v with heads merged back; w_out (optional) [H*B, Tq, Tk], head-major like the reference's
 * return_w (head h of user b at index h*B + b).  One wave per (user, head, query); any Tq, Tk. */
int carca_mha_core(const float* q, int ldq, const float* k, const float* v, int ldk, const int32_t* q_ids,
                   const int32_t* k_ids, int B, int Tq, int Tk, int d, int H, int has_causal, int causal, float* out,
                   int ldo, float* w_out /*or NULL*/, void* stream);
/* Which kernel each row product took.  carca_gemm_rows_log(NULL, 1) clears the calling thread's log and switches it on
 * ((NULL, 0): off); from then on every row-GEMM launch of that thread appends "kernel rows=.. N=.. K=.. grid=..;".
 * carca_gemm_rows_log(buf, cap) copies the log (NUL-terminated) and returns its length.  tools/bench_configs.py names each
 * configuration's dominant kernel with it. */
int carca_gemm_rows_log(char* out /*or NULL*/, int cap);
/* carca_mha_core with nn.Dropout on the weights (carca.py:258): W * keep / (1 - p) multiplies v, w_out stays pre-dropout
 * (carca.py:262-263); element (b, h, t, j) of site drop->site, keep-mask written to keep_out [B, H, Tq, Tk] (uint8,
 * or NULL).  drop NULL or p = 0: carca_mha_core.  The attention of profiles longer than the fused kernels' 64 slots
 * (carca_replication_amd/long_profile.py). */
int carca_mha_core_drop(const float* q, int ldq, const float* k, const float* v, int ldk, const int32_t* q_ids,
                        const int32_t* k_ids, int B, int Tq, int Tk, int d, int H, int has_causal, int causal, float* out,
                        int ldo, float* w_out /*or NULL*/, const CarcaDropout* drop /*or NULL*/, uint8_t* keep_out /*or NULL*/,
                        void* stream);
/* Backward of carca_mha_core (torch.autograd over carca.py:242-260): d_out [B*Tq, ldo] (or NULL) and d_w [H*B, Tq, Tk]
 * (or NULL: the gradient of the returned weights) -> dq [B*Tq, ldq] (written), dk, dv [B*Tk, ldk] (ACCUMULATED: zero
 * them first).  The weights are recomputed from q, k and the masks. */
int carca_mha_core_bwd(const float* q, int ldq, const float* k, const float* v, int ldk, const int32_t* q_ids,
                       const int32_t* k_ids, int B, int Tq, int Tk, int d, int H, int has_causal, int causal,
                       const float* d_out /*or NULL*/, int ldo, const float* d_w /*or NULL*/, float* dq, float* dk, float* dv,
                       void* stream);
/* carca_mha_core_bwd behind carca_mha_core_drop: keep = the forward's keep-mask [B, H, Tq, Tk] (NULL = no dropout),
 * keep_scale = 1 / (1 - p). */
int carca_mha_core_bwd_drop(const float* q, int ldq, const float* k, const float* v, int ldk, const int32_t* q_ids,
                            const int32_t* k_ids, int B, int Tq, int Tk, int d, int H, int has_causal, int causal,
                            const float* d_out /*or NULL*/, int ldo, const float* d_w /*or NULL*/, float* dq, float* dk,
                            float* dv, const uint8_t* keep /*or NULL*/, float keep_scale, void* stream);
/* Inverse of carca_pack_weights for gradients: real[r][c] (+)= packed[rp][cp] (same descriptor fields:
 * src = real tensor, dst = packed buffer).  accumulate = 0 overwrites, 1 adds. */
int carca_unpack_grads(const CarcaPackDesc* descs, int n, int accumulate, void* stream);

/* ---- backward of whole modules as ONE host call each (csrc/block_bwd.hip) --------------------------------------
 * The launch sequence of a module's backward is issued from C, like carca_forward issues the inference forward: the
 * training step is otherwise bound by interpreter time per launch.  Each entry launches its input-gradient chain on
 * `stream` and APPENDS its weight-gradient products to the caller's host array `wgrads` (*n_wgrads is advanced); the
 * caller launches them together with carca_gemm_wgrad_group once the whole pass is issued -- until then the
 * workspace and every saved tensor must stay alive.
 * carca_sa_block_bwd: autograd of SelfAttentionBlock.forward (carca.py:297-318): given dL/d(output) it produces
 * dL/d(input) and five products (ffn_2, ffn_1, W_Q, W_K, W_V with their biases); LayerNorm gammas / betas are
 * accumulated directly.  All gradient buffers are ACCUMULATED into (caller zeroes). */
typedef struct CarcaSaBwdDesc {
  int32_t B, L, d, H, residual;
  float drop_p;           /* the block's dropout probability in the forward (0 = none: masks unused) */
  const int32_t* ids;     /* [B*L] */
  const float* dy;        /* [B*L, DPI] gradient of the block's output */
  const float *x_in, *qn, *qh, *kh, *vh, *r, *s2, *h1; /* the block's input + CarcaSaSave of the forward */
  const uint8_t *m_attn, *m_ffn2;                      /* keep-masks of the forward (drop_p > 0) */
  const float *wq_t, *wk_t, *wv_t; /* [DPI, DPO]: Bt[n = input feature][k = head-padded output feature] = W[k][n] */
  const float *w1_t, *w2_t;        /* [DPI, DPI] transposed ffn weights */
  const float *ln1_w, *ln2_w;      /* LayerNorm gammas [d] */
  float *g_w1, *g_b1, *g_w2, *g_b2;             /* d ffn_1 / ffn_2: [d, d] (row stride d), [d] */
  float *g_wq, *g_wk, *g_wv, *g_bq, *g_bk, *g_bv; /* head-padded staging: [DPO, d] (row stride d), [DPO] */
  float *g_ln1_w, *g_ln1_b, *g_ln2_w, *g_ln2_b; /* [d] */
  float* workspace;       /* carca_sa_block_bwd_workspace(B, L, d, H) floats, alive until the products have run */
  float* dx;              /* out [B*L, DPI] gradient of the block's input */
} CarcaSaBwdDesc;
size_t carca_sa_block_bwd_workspace(int B, int L, int d, int H);
int carca_sa_block_bwd(const CarcaSaBwdDesc* desc /*host*/, CarcaWgradDesc* wgrads /*host, room for 5 more*/,
                       int* n_wgrads, void* stream);

/* carca_cross_score_bwd: autograd of the final LayerNorm (carca.py:421) + CrossAttentionBlock.forward (carca.py:338-349)
 * over every target group: given dL/dy per group it produces dL/d(embedded targets) per group (masked like e * mask),
 * dL/d(encoder output) and up to four products (ffn head, W_Q over all groups, W_K, W_V). */
typedef struct CarcaCrossBwdIn {
  const float* qh;       /* [B*N, DPO] saved by carca_cross_score_fwd */
  const float* y;        /* [B*N] forward scores */
  const float* dy;       /* [B*N] incoming gradient */
  const int32_t* ids;    /* [B*N] */
  const float* o;        /* [B*N, DPI] embedded targets (the forward's input) */
  const uint8_t* m_attn; /* keep-mask of the forward (drop_p > 0) or NULL */
  float* de;             /* out [B*N, DPI]: gradient of the embedded targets */
  int32_t N;
  int32_t ld_y;          /* elements between users in y AND dy; 0 = N */
} CarcaCrossBwdIn;
typedef struct CarcaCrossBwdDesc {
  int32_t B, L, d, H, ngroups, residual, training;
  float drop_p;
  CarcaCrossBwdIn group[CARCA_MAX_GROUPS];
  const int32_t* p_ids;                /* [B*L] */
  const float *kh, *vh;                /* [B*L, DPO] saved */
  const float* p_normed;               /* [B*L, DPI] final-norm output saved by the forward */
  const float* enc_out;                /* [B*L, DPI] encoder output (the final norm's input) */
  const float *wq_t, *wk_t, *wv_t;     /* [DPI, DPO] transposed head-padded copies */
  const float* ffn_w_pad;              /* [DPO] decoder.ffn.weight, head-padded (CarcaCaWeights.ffn_w_pad) */
  const float* ffn_w;                  /* [d] decoder.ffn.weight as is */
  const float* norm_w;                 /* [d] final LayerNorm gamma; NULL = the stand-alone CrossAttentionBlock on an already
                                          normed profile: no LayerNorm stage (enc_out, g_norm_* unused, dx = d p_normed)
                                          and de is the gradient of the block's INPUT o, not masked by ids */
  float *g_ffn_w, *g_ffn_b;            /* [d], [1] */
  float* g_ffn_w_pad;                  /* [DPO] head-padded staging of the attention part of d ffn.weight */
  float *g_wq, *g_wk, *g_wv, *g_bq, *g_bk, *g_bv; /* head-padded staging: [DPO, d], [DPO] */
  float *g_norm_w, *g_norm_b;          /* [d] */
  float* workspace;                    /* carca_cross_score_bwd_workspace(...) floats, alive until the products ran */
  float* dx;                           /* out [B*L, DPI]: gradient of the encoder output */
} CarcaCrossBwdDesc;
size_t carca_cross_score_bwd_workspace(int B, int L, int d, int H, const int32_t* Ns /*host [ngroups]*/, int ngroups);
int carca_cross_score_bwd(const CarcaCrossBwdDesc* desc /*host*/, CarcaWgradDesc* wgrads /*host, room for 4 more*/,
                          int* n_wgrads, void* stream);

/* carca_embed_bwd: autograd of AllEmbedding.forward (carca.py:85-95) over all row segments, given dL/de per segment
 * (NOT yet masked: the e * mask of carca.py:94 is applied here): position-encoding gradient, d joint_embed, d [z ; q],
 * item-row scatter-add, d feats_embed.  Everything is launched here (these products feed each other). */
typedef struct CarcaEmbedBwdSeg {
  const float* de;           /* [rows, ld_de] */
  const int32_t* ids;        /* [rows] */
  const float* attrs;        /* [rows, n_attrs] (or a [B, T, n_attrs] view, see attrs_bstride) */
  const float* ctx;          /* [rows, n_ctx] */
  const float* attrs_table;  /* optional [n_items, n_attrs]: rows gathered by id instead of `attrs` */
  int64_t attrs_bstride, ctx_bstride;
  int32_t rows, T, attrs_table_rows;
  int32_t joint_only;        /* 1: this segment only enters d joint_embed here -- its d [z ; q], scatter-add and d feats_embed
                              * are another call's (the same pass's second stream, see skip_joint) */
} CarcaEmbedBwdSeg;
typedef struct CarcaEmbedBwdDesc {
  CarcaEmbedBwdSeg seg[CARCA_MAX_SEGS]; /* seg[0] = the profile (the only one with a position encoding) */
  int32_t nseg, d, g, n_attrs, n_ctx, ld_de, L;
  const float* zq;        /* [sum rows, d + g] saved by carca_embed_fwd */
  const float* joint_wt;  /* [d + g, ld_joint_wt]: joint_embed.weight transposed (Bt of the input-gradient product) */
  int32_t ld_joint_wt;
  float *g_items, *g_feats_w, *g_feats_b, *g_joint_w, *g_joint_b; /* accumulated into (caller zeroes) */
  float* g_pos;           /* [L, d] gradient of LearnableEncoding.encoding.weight, or NULL */
  float* workspace;       /* carca_embed_bwd_workspace(...) floats */
  void* ev_early;         /* optional hipEvent_t recorded on `stream` right BEFORE the last launch (d feats_embed, as long as
                           * the whole forward GEMM): behind it every other gradient of the model is final, so a gradient
                           * all-reduce of everything but feats_embed.{weight,bias} can start under that kernel */
  int32_t skip_joint;     /* 1: no d joint_embed in this call (a later call of the pass lists these segments joint_only):
                           * a backward pass may hand the target rows' share to a second stream as soon as their d e is
                           * final and keep the small d joint_embed product in ONE launch over all rows */
  int32_t only_joint;     /* 1: d joint_embed (and g_pos) over the listed segments and NOTHING else -- the counterpart of
                           * skip_joint: a pass whose streams each ran their own rows with skip_joint issues the one small
                           * product over all rows wherever it has room (every segment may be joint_only) */
  void* table_stream;     /* optional second hipStream_t: the row table of the d feats_embed kernel (a function of the ids
                           * alone, one block's latency chain) is built there, forked off `stream` at entry and joined in front
                           * of the kernel, beside the launches of this call instead of between them.  NULL = on `stream` */
} CarcaEmbedBwdDesc;
size_t carca_embed_bwd_workspace(const int32_t* rows /*host [nseg]*/, int nseg, int d, int g);
int carca_embed_bwd(const CarcaEmbedBwdDesc* desc /*host*/, void* stream);
/* 1 when this thread's last carca_embed_bwd call with an ev_early recorded it.  On a stream that is being captured the
 * record is an EXTERNAL event-record node (every replay records the event in front of its last kernel); a HIP runtime that
 * offers no way to add one leaves the event alone and this returns 0: reduce the early range behind the replay then. */
int carca_early_event_recorded(void);

/* ---- f3: the optimizer step of the train driver (training.py:174, train.py:96) -------------------------------
 * torch.optim.Adam's update (no amsgrad; weight_decay added to the gradient) for every tensor of the table in one
 * launch: g += wd*p; m += (1-b1)(g-m); v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
 * `tensors` is a HOST array of n entries (device pointers, n elements each, fp32); step = t >= 1.  The scalar
 * hyper-parameters arrive as doubles and are combined in double before one rounding to fp32, as torch does. */
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  /* Optional row mask of an embedding table [n / row_len, row_len]: rows whose byte is 0 are SKIPPED.  Exact as long as
   * the caller marks every row that ever received a non-zero gradient and weight_decay is 0: a row with g = m = v = 0
   * has an update of exactly 0 (m, v stay 0, p -= lr * 0 / (0 + eps)), so skipping it gives the bits of the dense
   * sweep without reading or writing its four arrays (1 M items x 128: 2.5 GB per step).  NULL = every row. */
  const uint8_t* row_mask;
  int64_t row_len;
} CarcaAdamTensor;
int carca_adam_step(const CarcaAdamTensor* tensors, int n, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int step, void* stream);
/* mask[ids[i]] = 1 for i < n (ids outside [0, n_rows) are ignored): the rows a batch touches, for CarcaAdamTensor.row_mask */
int carca_mark_rows(const int32_t* ids, int64_t n, uint8_t* mask, int64_t n_rows, void* stream);
/* table[ids[i]][0 .. row_len) = 0 for every id of up to CARCA_MAX_SEGS id lists (ids outside [0, n_rows) ignored): clears
 * the rows the previous step's scatter-add touched instead of the whole gradient table.  carca_concat_ids copies the
 * lists back to back into `out` (sum of counts entries): the caller's private record of a step's "dirty" rows. */
int carca_zero_rows(float* table, int64_t n_rows, int row_len, const int32_t* const* ids /*host [nlists]*/,
                    const int64_t* counts /*host [nlists]*/, int nlists, void* stream);
int carca_concat_ids(const int32_t* const* ids /*host [nlists]*/, const int64_t* counts /*host*/, int nlists,
                     int32_t* out, void* stream);

/* ---- f1: batch construction on the device (data.py:53-192) ---------------------------------------------------
 * The interaction log lives in HBM as CSR: user u owns hist[offs[u] .. offs[u+1]) (item ids in interaction order) and
 * the context rows hctx[offs[u] ..] of those interactions (hctx[pos] = ctx[(user, hist[pos])], data.py:17-25).
 * `users` selects the batch; held_out / floor_ are pad_profile's split constants (data.py:53-74):
 *   train: held_out = 2 if a test split exists else 1, floor_ = 1;  val: 1 or 0, floor_ = 2;  test: 0, floor_ = 3.
 * Negatives are distinct ids in [1, n_items-1] outside the user's whole history (data.py:77-87), drawn by rejection
 * from a counter-based hash of (seed, user, attempt): reproducible per seed, not python's `random` stream.
 * carca_build_eval_batch = get_test_sequences (data.py:140-192): p_x [B,L] left-padded history, p_c [B,L,n_ctx],
 *   o_x [B,1+N] = held-out item then N negatives, o_c [B,1+N,n_ctx] = the held-out interaction's context for every
 *   candidate, y_true [B,1+N] = one-hot at column 0.  N <= 2048.
 * carca_build_train_batch = get_train_sequences (data.py:90-137): o_x [B,2L] = successors | negatives aligned with the
 *   history slots, o_c [B,2L,n_ctx] = the successor's context for both, y_true [B,2L] = (p_x > 0) | 0.
 * Attribute rows are NOT materialised: AllEmbedding.register_attr_table gathers them inside the feature GEMM.
 * offs has n_users + 1 entries; a user index outside [0, n_users) yields an all-pad row (no history, no candidates). */
int carca_build_eval_batch(const int32_t* hist, const int64_t* offs, const float* hctx, const int32_t* users, int n_users,
                           int B, int L, int N, int n_ctx, int n_items, int held_out, int floor_, uint64_t seed, int32_t* p_x,
                           float* p_c, int32_t* o_x, float* o_c, int32_t* y_true, void* stream);
int carca_build_train_batch(const int32_t* hist, const int64_t* offs, const float* hctx, const int32_t* users, int n_users,
                            int B, int L, int n_ctx, int n_items, int held_out, int floor_, uint64_t seed, int32_t* p_x,
                            float* p_c, int32_t* o_x, float* o_c, int32_t* y_true, void* stream);

/* ---- a7: CARCA.forward (carca.py:411-431), inference path, as ONE host call -----------------------------
 * Issues the whole launch sequence -- gather, feature GEMM, joint GEMM, every SelfAttentionBlock, final norm +
 * grouped cross-attention scoring -- on `stream` without returning to the caller in between, so that a slow
 * host thread cannot starve the GPU (the Python layer needs ~25 interpreter-level calls per forward otherwise).
 * No tensors are saved for a backward pass and no dropout is applied: eval mode, or train mode without grad
 * at p = 0.  All buffers are caller-allocated:
 *   segs[0] = profile, segs[1..ngroups] = target groups (their e_out feed the scoring kernel),
 *   x_work[2] = two [B*L, ld_e] ping-pong buffers for the blocks' outputs.
 * ev (optional, host array of 4 hipEvent_t, entries may be NULL): so that a benchmark can time exactly these kernels
 * inside its timed region.  ev[0], ev[1] (both or neither) are BOUND to the feature GEMM's dispatch on `stream`
 * (hipExtLaunchKernel: start / end of that kernel, hipEventElapsedTime(ev[0], ev[1]) = its duration; nothing extra is
 * queued).  ev[2], ev[3] (both or neither) are bound the same way to the scoring kernel's dispatch (a hipEventRecord
 * would be a barrier packet of its own: ~6 us of GPU time between two kernels). */
#define CARCA_MAX_BLOCKS 8
typedef struct CarcaForwardDesc {
  CarcaRowSeg segs[CARCA_MAX_SEGS];
  int32_t ngroups;
  int32_t B, L, d, g, H, n_attrs, n_ctx, n_blocks, ld_e;
  const float *items_w, *feats_w, *feats_b, *joint_w, *joint_b, *pos;
  float* zq;
  float* x_work[2];
  CarcaSaWeights sa[CARCA_MAX_BLOCKS];
  int32_t sa_residual[CARCA_MAX_BLOCKS];
  CarcaCaWeights ca;
  int32_t ca_residual, training;
  float* y[CARCA_MAX_GROUPS];   /* [B, N_g] outputs, row stride ldy */
  int32_t N[CARCA_MAX_GROUPS];
  int32_t ldy;                  /* 0 = every group dense [B, N_g]; else the groups are column blocks of one [B, ldy] tensor */
  float* p_normed;              /* optional [B*L, ld_e] */
  /* Optional FOLDED embedding (inference with frozen weights).  AllEmbedding has no nonlinearity (carca.py:86-89),
   * so e = z W_jz^T + [a;c] (W_jq W_f)^T + (W_jq b_f + b_j): with fold_wc = W_jq W_f [d, F] (row stride
   * fold_ldwc) and fold_bias [d] prepared once per weight version, the F->g->d pair of GEMMs becomes one F->d
   * GEMM (5x fewer executed flops at g = 5d) and zq is not touched.  Same algebra, different fp32 summation
   * order (~1e-6 relative); NULL = the two-GEMM path that mirrors the reference operation by operation. */
  const float* fold_wc;
  const float* fold_bias;
  int32_t fold_ldwc;
  /* TRAINING extras (all zero / NULL = the inference forward): the same launch sequence with the tensors the backward
   * entries (carca_sa_block_bwd, carca_cross_score_bwd, carca_embed_bwd) need, and the reference's dropout sites.
   *   x_out[i]     block i writes its output here instead of the ping-pong buffers (block i+1's input is saved)
   *   sa_save[i]   CarcaSaSave of block i (save_blocks = 1);  ca_save: CarcaCaSave of the scoring kernel (save_cross = 1);
   *                p_normed (above) receives the final norm's output
   *   p_embed / p_block / p_cross  dropout probabilities of CARCA.dropout (carca.py:416), of every block's three sites
   *                and of the decoder's attention weights; one `seed` per forward; sites 1000, 4 i .. 4 i + 2, 2000 + g;
   *                m_embed [B*L, d] receives the embedding dropout's keep-mask */
  float* x_out[CARCA_MAX_BLOCKS];
  CarcaSaSave sa_save[CARCA_MAX_BLOCKS];
  CarcaCaSave ca_save;
  int32_t save_blocks, save_cross;
  float p_embed, p_block, p_cross;
  uint64_t seed;
  uint8_t* m_embed;
  const uint64_t* seed_offset; /* device or NULL, see CarcaDropout */
  int32_t n_events;            /* entries of carca_forward's `ev` array: 0 or 4 = the four described there, 8 = four more:
                                * ev[4], ev[5] bound to the FIRST SelfAttentionBlock's dispatch, ev[6], ev[7] to the joint
                                * GEMM's (AllEmbedding.joint_embed, carca.py:89) */
  /* Optional PROJECTED item table (inference, nothing saved for a backward): z_table[i] = sqrt(d) items_w[i] W_jz^T,
   * [n_items, ld_z_table], prepared once per weight version.  joint_embed is linear (carca.py:89), so
   *   e = [z ; q] W_j^T + b_j = q W_jq^T + z_table[id] + b_j:
   * the item rows are never gathered into zq (no gather launch, carca.py:87-88), the joint product runs over the g
   * columns of q only, and the item term is added per row in its epilogue.  Same algebra, another fp32 summation order
   * (~1e-7 relative).  NULL = the reference's operation order (gather, then one product over d + g columns). */
  const float* z_table;
  int32_t ld_z_table;
} CarcaForwardDesc;
int carca_forward(const CarcaForwardDesc* desc /*host*/, void* const* ev /*4 hipEvent_t or NULL*/, void* stream);
/* Event helpers so that a host language without a HIP binding can time kernels on the launch stream. */
int carca_event_create(void** ev_out);
int carca_event_destroy(void* ev);
int carca_event_elapsed_ms(void* start, void* stop, float* ms_out); /* both must have completed */
int carca_stream_wait_event(void* stream, void* event);            /* hipStreamWaitEvent */

/* ---- a8: BinaryCrossEntropy.forward (carca.py:441-444) -----------------------------------------
 * loss = sum(l * m) / sum(m), l = -(t log(y+eps) + (1-t) log(1-y+eps)), m = (ids != 0).
 * scratch: 2 floats, written by the call as [sum(l*m), sum(m)]; loss_out: 1 float.  dy (optional)
 * receives dloss/dy.  denom (optional, device float[1]) replaces sum(m) as the normaliser: with users
 * sharded over ranks it holds the all-reduced mask count, so that summing the ranks' gradients gives
 * exactly the single-process batch gradient. */
int carca_bce_fwd(const float* y, const int32_t* y_true, const int32_t* ids, int n, float eps, float* scratch,
                  float* loss_out, float* dy /*or NULL*/, const float* denom /*or NULL*/, void* stream);

/* ---- M: compute_HR / compute_NDCG (train.py:15-32) without the sort ----------------------------
 * With p = pos[u] (the positive's column; NULL = column 0 as in data.py:165,190):
 * rank[u] = #{j != p : y[u][j] > y[u][p]}; sums[0] += [rank < k], sums[1] += [rank<k]/log2(rank+2),
 * sums[2] += #{j != p : y[u][j] == y[u][p]} (ties: the reference's unstable sort is undefined there).
 * sums is accumulated into (caller zeroes it once per evaluation). */
/* One evaluation batch of train.py:41-51 in one launch: sums[0..2] as carca_rank_metrics with the positive in column 0
 * (data.py:165,190), sums[3] += BinaryCrossEntropy(y, y_true, ids != 0) (carca.py:441-444), sums[4] += B; fixed summation
 * order (reproducible).  B x N <= 16384, else CARCA_ERR_UNSUPPORTED (use the two calls). */
int carca_eval_metrics(const float* y /*[B,N]*/, const int32_t* y_true, const int32_t* ids, int B, int N, int k, float eps,
                       float* sums /*[5]*/, void* stream);
int carca_rank_metrics(const float* y /*[B,N]*/, int B, int N, int k, const int32_t* pos /*[B] or NULL*/,
                       int32_t* rank /*[B] or NULL*/, float* sums /*[3]*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CARCA_HIP_H */
