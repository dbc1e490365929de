/*
 * gctplus_hip.h -- C ABI of libgctplus_hip.so (gfx950 / MI355X).
 *
 * The reference (chaoting-sun/GCT-Plus) is pure Python/PyTorch and has NO plugin,
 * operator or FFI interface; the drop-in boundary is the Python
 * nn.Module contract (Model/build_model.py:79-87 get_model, Vaetf/Cvaetf.forward
 * Model/vaetf.py:154-182, Model/cvaetf.py:179-193).  This header is the ABI the
 * build adds UNDERNEATH that contract: one entry point per ATen call site group of
 * SURVEY.md 2.2 (K1..K10).  Each declaration cites the reference call site whose
 * arithmetic it replaces.
 *
 * Conventions
 *  - plain C: raw device pointers, sizes, scalars, a hipStream_t passed as void*.
 *  - all tensors fp32 row-major unless stated; token ids int64; masks uint8.
 *  - the caller owns every buffer (PyTorch caching allocator); the library allocates
 *    nothing and never synchronises; kernels are enqueued on `stream`.
 *  - return 0 on success, <0 on error (GCT_ERR_*); message via gct_last_error()
 *    (thread-local).  Never throws, never exits.
 *  - "segmented" matrices: a logical [R][nseg*nper] matrix whose column block s
 *    (or row block s for weights) lives behind its own pointer p[s].  This is how the
 *    separate q/k/v (and mu/log_var) nn.Linear parameters of the reference are fused
 *    into one GEMM without repacking or renaming any checkpoint tensor.
 *  - dropout: Philox4x32-10, key = (seed, site), counter = element coordinates; the
 *    backward kernels regenerate the mask from (seed, site), nothing is stored.
 *    p == 0 disables it (bit-exact parity mode, SURVEY.md 7 "RNG").
 */
#ifndef GCTPLUS_HIP_H
#define GCTPLUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCT_OK 0
#define GCT_ERR_ARG (-1)  /* bad shape / alignment / null pointer            */
#define GCT_ERR_HIP (-2)  /* a HIP runtime call failed (launch error)        */

#define GCT_ABI_VERSION 13

int gct_version(void);
const char* gct_last_error(void);

/* workspace sizing helpers (bytes); the caller allocates */
int64_t gct_wgrad_ws_bytes(int64_t M, int64_t Ntot, int64_t K);
int64_t gct_rowred_ws_bytes(int64_t rows, int64_t cols);

/* ------------------------------------------------------------------ K2: Norm */
/* Model/modules.py:92-95   y = alpha*(x-mean)/(std_unbiased+eps)+bias.
 * Saves mean[rows] and rstd[rows] = 1/(std+eps) for the backward. d % 4 == 0, d <= 2048. */
int gct_norm_fwd(const float* x, const float* alpha, const float* bias, float* y,
                 float* mean, float* rstd, int64_t rows, int d, float eps, void* stream);
/* dx = dNorm/dx (dy) [+ dres]; dalpha, dbias overwritten. ws >= gct_rowred_ws_bytes(rows, 2*d).
 * quad_map (nullable): dy / dres / dx are QUAD-COMPACTED rows (see gct_live_rows) while x, mean, rstd stay in the
 * forward's row space of src_rows rows: compact row 4i+e reads x row 4*quad_map[i]+e. */
int gct_norm_bwd(const float* dy, const float* x, const float* alpha, const float* mean,
                 const float* rstd, const float* dres, float* dx, float* dalpha, float* dbias,
                 float* ws, int64_t rows, int d, float eps, const int32_t* quad_map, int64_t src_rows,
                 float* drop_out, float p, uint64_t seed, uint32_t site, void* stream);
/* drop_out (nullable, same rows as dx): additionally dropout_bwd(dx) with the mask (seed, site, p) of the dropout the
 * preceding sub-layer applied to its output -- that sub-layer's backward reads it instead of running gct_dropout_bwd. */

/* --------------------------------------------------- K1: embedding + PE (+cond) */
/* Model/modules.py:108-110 (lookup), :134-144 (x*sqrt(d)+pe, dropout),
 * Model/vaetf.py:35-39 (cond rows concatenated in front).  out row (b,l):
 *   l <  n_c : cond[b][l][:]        (cond = embed_cond2enc(econds) viewed [B,n_c,d])
 *   l >= n_c : table[tok[b][l-n_c]][:]
 * then *scale + pe[l][:], dropout(site).  L = n_c + S. */
int gct_embed_pe_fwd(const int64_t* tok, const float* table, const float* cond, const float* pe,
                     float* out, int B, int S, int n_c, int d, int vocab, float scale, float p,
                     uint64_t seed, uint32_t site, void* stream);
/* dtable[vocab][d] overwritten (deterministic two-stage reduction), dcond[B][n_c][d]
 * overwritten (nullable when n_c == 0). ws >= gct_embed_ws_bytes. */
int64_t gct_embed_ws_bytes(int B, int S, int d, int vocab);
int gct_embed_pe_bwd(const float* dout, const int64_t* tok, float* dtable, float* dcond,
                     float* ws, int B, int S, int n_c, int d, int vocab, float scale, float p,
                     uint64_t seed, uint32_t site, void* stream);

/* ------------------------------------------------------------- K3: nn.Linear */
/* Model/sublayers.py:54-59,64-66,70 (q/k/v/out), :81-88 (FFN), :11-12 (fc_mu,
 * fc_log_var), Model/vaetf.py:81 (fc_z), :133/:169 (out).  fp32-input MFMA
 * (v_mfma_f32_32x32x2_f32): results are exact-fp32 fma chains.
 *
 * y_s[m][n] = epi( sum_k x[m][k] * w_s[n][k] + b_s[n] )   s = 0..nseg-1, n < nper
 *   GCT_EPI_BIAS       : plain
 *   GCT_EPI_GELU_DROP  : pre[m][n] = acc+b (saved), y = dropout(gelu_erf(pre))   (FFN-1)
 *   GCT_EPI_DROP_RESID : y = resid[m][n] + dropout(acc+b)                         (out / FFN-2)
 * pre/resid share y's leading dimension and are only valid with nseg == 1. */
#define GCT_EPI_BIAS 0
#define GCT_EPI_GELU_DROP 1
#define GCT_EPI_DROP_RESID 2
int gct_linear_fwd(const float* x, int64_t ldx, int64_t M, int K,
                   const float* w0, const float* w1, const float* w2, int64_t ldw,
                   const float* b0, const float* b1, const float* b2, int nseg, int nper,
                   float* y0, float* y1, float* y2, int64_t ldy,
                   int epi, const float* resid, float* pre, float p, uint64_t seed, uint32_t site,
                   void* stream);

/* Same contract with a caller-provided workspace (>= gct_linear_fwd_ws_bytes): lets skinny-M
 * shapes (decode steps, M = batch rows) run split-K with a fused reduce+epilogue pass so all CUs
 * get work; identical results up to fp32 summation order. */
int64_t gct_linear_fwd_ws_bytes(int64_t M, int K, int Ntot);
int gct_linear_fwd_ws(const float* x, int64_t ldx, int64_t M, int K,
                      const float* w0, const float* w1, const float* w2, int64_t ldw,
                      const float* b0, const float* b1, const float* b2, int nseg, int nper,
                      float* y0, float* y1, float* y2, int64_t ldy,
                      int epi, const float* resid, float* pre, float p, uint64_t seed, uint32_t site,
                      float* ws, int64_t ws_bytes, void* stream);
/* ws_bytes: size of ws.  A route that needs more slabs than ws holds takes fewer K-splits (same result up to the
 * fp32 summation order) or no workspace route at all; nothing is ever written past ws + ws_bytes. */

/* dx[m][k] (op)= sum_s sum_n dy_s[m][n] * w_s[n][k]
 *   GCT_DEPI_STORE / GCT_DEPI_ACCUM (dx += ...) /
 *   GCT_DEPI_GELU_BWD : dx = acc * gelu'(pre[m][k]) * dropmask(site)/(1-p)   (through FFN-1 act.) */
#define GCT_DEPI_STORE 0
#define GCT_DEPI_ACCUM 1
#define GCT_DEPI_GELU_BWD 2
int gct_linear_dgrad(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                     int64_t M, int nseg, int nper,
                     const float* w0, const float* w1, const float* w2, int64_t ldw, int K,
                     float* dx, int64_t lddx, int depi, const float* pre, float p, uint64_t seed,
                     uint32_t site, void* stream);

/* dw_s[n][k] = sum_m dy_s[m][n] * x[m][k] ; db_s[n] = sum_m dy_s[m][n]  (overwrite).
 * Split over M into fp32 slabs in ws, reduced deterministically (no atomics).
 * ws >= gct_wgrad_ws_bytes(M, nseg*nper, K). db* nullable. */
int gct_linear_wgrad(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                     int64_t M, int nseg, int nper, const float* x, int64_t ldx, int K,
                     float* dw0, float* dw1, float* dw2, int64_t lddw,
                     float* db0, float* db1, float* db2, float* ws, void* stream);

/* ---- GEMM arithmetic mode and pre-split weights ------------------------------------------
 * The nn.Linear GEMMs (Model/sublayers.py:54-59,64-66,70,81-88) run in one of two modes, both with
 * fp32 operands, fp32 results and fp32-class error:
 *   GCT_GEMM_F32    : v_mfma_f32_32x32x2_f32 (fp32 fma chains);
 *   GCT_GEMM_BF16X6 : every operand element is split EXACTLY into three bf16 values (8+8+8
 *                     significand bits) and the six leading partial products are accumulated in
 *                     fp32 by v_mfma_f32_16x16x32_bf16; launches that do not qualify (odd shapes,
 *                     skinny M, no planes for fwd/dgrad) use the fp32 kernels.
 * Process-wide; default from the environment (GCT_GEMM_MODE=f32|x6, x6 when unset). */
#define GCT_GEMM_F32 0
#define GCT_GEMM_BF16X6 1
int gct_gemm_set_mode(int mode);
int gct_gemm_get_mode(void);
/* diagnostics: out2[0] = launches of the fp32-MFMA tile kernels so far, out2[1] = of the bf16x6 kernels */
int gct_gemm_launch_counts(int64_t* out2);
/* diagnostics: gemm_x6_kernel launches so far (a tail-balanced forward / dgrad call launches it twice) */
int64_t gct_gemm_x6_kernel_launches(void);

/* planes[p*plane_stride + i] = p-th bf16 piece (p = 0 high, 1 middle, 2 low) of src[i], i < numel.
 * numel % 4 == 0, plane_stride % 4 == 0, plane_stride >= numel; src 16-B aligned.  Run it over the
 * model's flat parameter buffer once per step: the plane of a weight that starts at element offset o
 * of the buffer starts at element offset o of every plane. */
int gct_split_planes(const float* src, int64_t numel, uint16_t* planes, int64_t plane_stride,
                     void* stream);

/* gct_linear_fwd_ws / gct_linear_dgrad with the weights' bf16 planes: wp0 = plane 0 of w0; the planes
 * of w1, w2 are addressed as wp0 + (w1 - w0), wp0 + (w2 - w0) (the layout gct_split_planes produces
 * over a common buffer).  wp0 == NULL, or mode GCT_GEMM_F32: identical to the plain entry points. */
int gct_linear_fwd_p(const float* x, int64_t ldx, int64_t M, int K,
                     const float* w0, const float* w1, const float* w2, int64_t ldw,
                     const uint16_t* wp0, int64_t plane_stride,
                     const float* b0, const float* b1, const float* b2,
                     int nseg, int nper, float* y0, float* y1, float* y2, int64_t ldy,
                     int epi, const float* resid, float* pre, float p, uint64_t seed, uint32_t site,
                     float* ws, int64_t ws_bytes, const int32_t* quad_map, void* stream);
/* quad_map of gct_linear_fwd_p (nullable): the M rows are a quad compaction of a larger row space (gct_live_rows); the
 * dropout masks of the GCT_EPI_GELU_DROP / GCT_EPI_DROP_RESID epilogues are then drawn at the coordinates of original quad
 * quad_map[q] for compact quad q, i.e. every live row gets exactly the mask it would get in the full row space. */
int gct_linear_dgrad_p(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                       int64_t M, int nseg, int nper,
                       const float* w0, const float* w1, const float* w2, int64_t ldw,
                       const uint16_t* wp0, int64_t plane_stride, int K,
                       float* dx, int64_t lddx, int depi, const float* pre, float p, uint64_t seed,
                       uint32_t site, float* ws, int64_t ws_bytes, const int32_t* quad_map, int64_t pre_rows,
                       void* stream);
/* quad_map (nullable): the M rows are quad-compacted (gct_live_rows); GCT_DEPI_GELU_BWD then regenerates the dropout
 * mask of compact quad q from original quad quad_map[q].  pre_rows == 0: pre is compact like dy / dx; pre_rows > 0:
 * pre keeps the forward's row space (pre_rows rows) and is read through the quad map (no gathered copy needed). */
/* ws of gct_linear_dgrad_p (nullable): >= gct_linear_dgrad_ws_bytes(M, nseg*nper, K).  With a workspace the
 * bf16x6 forward / dgrad launches balance a partial last round of tiles (the tail rows as a second launch: on 64 x 128
 * tiles when the reduction is <= 1024 long, else K-split into slabs + a fix-up kernel) and split a long reduction over
 * few tiles across the whole chip; gct_linear_fwd_ws_bytes covers the forward (skinny split-K or tail slabs, whichever
 * the launch would use).  Without one every launch is a single kernel. */
int64_t gct_linear_dgrad_ws_bytes(int64_t M, int Ntot, int K);

/* Zero-gradient rows.  Under an ignore_index loss the rows of padded target positions carry exactly zero
 * gradients through the whole decoder backward.  gct_nonzero_row_tiles lists, in ascending order, the 32-row
 * tiles of x[rows][cols] that hold a non-zero element (list: >= ceil(rows/32) int32; count: device scalar;
 * flags_ws: >= ceil(rows/32) bytes); gct_linear_wgrad_kt is gct_linear_wgrad reducing only over the listed
 * token tiles -- exact whenever every dy row outside them is zero (the skipped terms are 0 * x).  The list is
 * honoured by the bf16x6 kernel; the fp32 kernels reduce over all rows (same result). */
/* *counter += number of rows of g[rows][cols] with live[row] == 0 that hold a non-zero element: the check a forward
 * that SKIPPED the dead rows (decoder forward over the rows that reach the loss) owes its backward -- a gradient on a
 * skipped row cannot be honoured.  The caller reads the counter at its next host synchronisation. */
int gct_dead_rows_nonzero(const float* g, int64_t ld, int64_t rows, int cols, const uint8_t* live,
                          int32_t* counter, void* stream);
int gct_nonzero_row_tiles(const float* x, int64_t ld, int64_t rows, int cols, int32_t* list,
                          int32_t* count, uint8_t* flags_ws, void* stream);
/* The property above holds only if no live query row attends to a dead (zero-gradient) row: a dead row that is a
 * VISIBLE KEY of a live query receives dK / dV (and a live row that sees NO key attends uniformly to all of them).
 * gct_live_rows derives the live rows of g[B*T][cols] and CHECKS the property on the device against the
 * self-attention mask actually used (mask element (b,i,j) at mask[b*mask_sb + i*mask_sq + j]; NULL = everything
 * visible; mask_sq == 0 = key-padding mask):
 *   live [B*T] u8; n_b [B] live rows per sample;
 *   info [8] i32: [0] live rows, [1] samples that violate the property, [2] samples whose live rows are not the
 *                 prefix 0..n_b-1, [3] listed token tiles, [4] compact rows (multiple of 128), [5] live quads;
 *   tile_list / tile_count / tile_flags_ws (nullable, together): the 32-row token tiles that hold a live row --
 *   EVERY tile when info[1] != 0, so gct_linear_wgrad_kt stays exact without a host round trip;
 *   cstart [B] / quad_list [ceil(B*T/4)+32] / qrank_ws [ceil(B*T/4)] (nullable, together): the COMPACTION MAP of the
 *   decoder backward.  Rows are compacted in aligned groups of 4 ("quads"), the granularity at which every dropout
 *   site draws its Philox values: compact row 4i+e <-> original row 4*quad_list[i]+e (quad_list ascending, padded
 *   with -1 to a multiple of 32 quads); sample b's row t sits at compact row cstart[b]+t.
 * gct_gather_quads / gct_scatter_quads move rows between the two spaces (scatter: dst pre-zeroed by the caller). */
int gct_live_rows(const float* g, int64_t ld, int B, int T, int cols, const uint8_t* mask, int64_t mask_sb,
                  int64_t mask_sq, uint8_t* live, int32_t* n_b, int32_t* info, int32_t* cstart,
                  int32_t* quad_list, int32_t* qrank_ws, int32_t* tile_list, int32_t* tile_count,
                  uint8_t* tile_flags_ws, void* stream);
/* The same compaction map for the KEY side of cross-attention, from a key-padding mask [B][Lk] (element (b,k) at
 * mask[b*mask_sb + k]): padded rows of the encoder memory are masked keys -- their K / V projections are never used
 * and their dK / dV are zero -- so the K / V GEMMs and their backward can run on the visible rows only.  Usable when
 * info[2] == 0 (visible keys are a prefix of every row) and info[6] == 0 (every sample sees a key).
 * info: [0] visible keys, [2] non-prefix samples, [4] compact rows, [5] live quads, [6] samples without a visible key. */
int gct_key_rows(const uint8_t* mask, int64_t mask_sb, int B, int Lk, uint8_t* live, int32_t* n_b, int32_t* info,
                 int32_t* cstart, int32_t* quad_list, int32_t* qrank_ws, void* stream);
int gct_gather_quads(const float* src, int64_t ld, int64_t M, const int32_t* quad_list, int64_t nrows, int cols,
                     float* dst, int64_t ldd, void* stream);
int gct_scatter_quads(const float* src, int64_t ld, const int32_t* quad_list, int64_t nrows, int cols, float* dst,
                      int64_t ldd, int64_t M, void* stream);
/* Zero the rows of a compact [nrows][cols] buffer that belong to no sample: [cstart[b] + n_b[b], cstart[b+1]) for every
 * b and [.., nrows) behind the last one -- what a kernel that writes only rows cstart[b] .. + n_b[b] (attention over
 * compact rows) leaves untouched.  cstart must be ascending (gct_live_rows / gct_key_rows). */
int gct_zero_gap_rows(float* buf, int64_t ld, int cols, const int32_t* cstart, const int32_t* n_b, int B, int64_t nrows,
                      void* stream);
/* dst rows += the compact rows (gradient of rows that were gathered: the encoder's K | V over its visible rows) */
int gct_scatter_add_quads(const float* src, int64_t ld, const int32_t* quad_list, int64_t nrows, int cols, float* dst,
                          int64_t ldd, int64_t M, void* stream);
int gct_linear_wgrad_kt(const float* dy0, const float* dy1, const float* dy2, int64_t lddy,
                        int64_t M, int nseg, int nper, const float* x, int64_t ldx, int K,
                        float* dw0, float* dw1, float* dw2, int64_t lddw,
                        float* db0, float* db1, float* db2, float* ws,
                        const int32_t* kt_list, const int32_t* kt_count, void* stream);

/* elementwise dropout backward for the GCT_EPI_DROP_RESID sites: dy = dropmask*dout/(1-p); quad_map (nullable):
 * the rows are quad-compacted, the mask of compact quad q is that of original quad quad_map[q] */
int gct_dropout_bwd(const float* dout, float* dy, int64_t rows, int cols, float p, uint64_t seed,
                    uint32_t site, const int32_t* quad_map, void* stream);

/* ------------------------------------------------------------- K4: attention */
/* Model/sublayers.py:29-41 attention() + head split/merge :64-69.
 * q,k,v are [B][L][H*dk]-shaped views with leading dimension ld* (so the fused QKV
 * buffer is consumed in place); head h uses columns h*dk..h*dk+dk-1.
 * mask: PACKED by gct_attn_mask_pack -- one bit per key, 8 uint32 words per query row (Lk <= 256), bit (k & 31)
 * of word k >> 5 set <=> key k visible; row of (b,q) at mbits + b*mb_sb + q*mb_sq (strides in words, multiples of
 * 4; mb_sq == 0 broadcasts a key-padding mask); a cleared bit => score := -1e9 (masked_fill semantics). nullable.
 * Pack once per forward: the same rows serve every layer, every head and the backward pass.
 * o: [B][Lq][H*dk] (heads merged, ready for the out projection); lse: [B][H][Lq].
 * probs (nullable): pre-dropout probabilities [B][H][Lq][Lk] (get_attn path).
 * dk in {16, 32, 64}; Lq, Lk <= 208 (the reference's positional table ends at 200: Model/modules.py:117; + 3
 * condition tokens).  Lk <= 96 runs the barrier-free kernels (one wave per query / key tile), longer rows the LDS kernels. */
/* mask: uint8, element (b,q,k) at mask[b*mask_sb + q*mask_sq + k] (0 = masked); mask_sq == 0: key-padding mask
 * [B][Lk] -> bits [B][8]; else bits [B][Lq][8]. */
/* tiles (nullable): one word per (batch, 16-row query tile) -- [B][1] for a key-padding mask, [B][ceil(Lq/16)]
 * otherwise -- bit t set <=> key tile t must be visited; passed to gct_attn_fwd / gct_attn_bwd as tbits with the
 * strides (tb_sb, tb_su) = (1, 0) resp. (ceil(Lq/16), 1), it lets a wave request its K rows before its mask rows
 * are back.  The kernels compute the same word themselves when tbits is NULL. */
int gct_attn_mask_pack(const uint8_t* mask, int64_t mask_sb, int64_t mask_sq, int B, int Lq, int Lk,
                       uint32_t* bits, uint32_t* tiles, void* stream);
/* The decoder's self-attention mask straight from the token ids: the uint8 form of get_trg_mask(target, pad, False)
 * (Model/modules.py:17-30, 47-58: pad mask & no-peek pattern * pad_idx, so out[b][q][k] = (tokens[b][k] != pad) &
 * (k <= q) & (pad & 1)) in ONE launch instead of the reference's int64 [B][T][T] tensor and its ~10 elementwise
 * launches.  tokens: [B] rows of T ids, leading dimension ld_tok; out: uint8 [B][T][T], 4-byte aligned. */
int gct_trg_mask_tokens(const int64_t* tokens, int64_t ld_tok, int64_t pad, int B, int T, uint8_t* out, void* stream);
int gct_attn_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                 int64_t ldv, const uint32_t* mbits, int64_t mb_sb, int64_t mb_sq,
                 float* o, int64_t ldo, float* lse, float* probs, int B, int H, int Lq, int Lk,
                 int dk, float scale, float p, uint64_t seed, uint32_t site, const int32_t* kstart,
                 const int32_t* klen, const uint32_t* tbits, int64_t tb_sb, int64_t tb_su,
                 const int32_t* qstart, const int32_t* qlen, void* stream);
/* kstart / klen (nullable, together; gct_key_rows): k and v hold only the VISIBLE keys of every sample, quad-compacted:
 * the rows of sample b start at kstart[b] and there are klen[b] of them (keys klen[b]..Lk-1 are masked by mbits).
 * qstart / qlen (nullable, together; gct_live_rows): q and o hold only the first qlen[b] query rows of sample b, at rows
 * qstart[b].. (the decoder forward over the rows that reach the loss); lse keeps its [B][H][Lq] layout, entries of rows
 * that do not exist are not written.  Needs the direct kernels (Lk <= 96, probs == NULL). */
/* dq/dk/dv written (overwrite) with the same layout as q/k/v.
 * cstart / nlive (nullable, together): dout and dq are quad-compacted (gct_live_rows): the rows of sample b start at
 * cstart[b] and only its first nlive[b] query rows exist; kv_compact bit 0 (self-attention, Lq == Lk): dk / dv live
 * in those compact rows too while q, k, v, o stay in the forward's layout; kv_compact bit 1: q and o are compact like
 * dout / dq (the forward ran with qstart / qlen; pass kstart / klen = cstart / nlive for self-attention) -- direct
 * kernels only.  lse keeps the forward's layout. */
int gct_attn_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                 int64_t ldv, const uint32_t* mbits, int64_t mb_sb, int64_t mb_sq,
                 const float* o, const float* dout, int64_t ldo, const float* lse,
                 float* dq, int64_t lddq, float* dk_, int64_t lddk, float* dv, int64_t lddv,
                 int B, int H, int Lq, int Lk, int dk, float scale, float p, uint64_t seed,
                 uint32_t site, const int32_t* cstart, const int32_t* nlive, int kv_compact,
                 const int32_t* kstart, const int32_t* klen, const uint32_t* tbits, int64_t tb_sb, int64_t tb_su,
                 void* ws, int64_t ws_bytes, void* stream);
/* kstart / klen as in gct_attn_fwd: k, v AND dk, dv hold the visible keys only (excludes kv_compact).
 * ws (nullable, caller-owned, 16-B aligned, >= gct_attn_bwd_ws_bytes): scratch of the two-launch backward used for
 * Lk <= 96 (one wave per query tile -> dq, then one wave per key tile -> dk, dv; no LDS); without it, or beyond 96
 * keys, the single-launch LDS kernel runs.  Same results either way (same arithmetic, same dropout bits). */
int64_t gct_attn_bwd_ws_bytes(int B, int H, int Lq, int Lk);

/* ------------------------------------------------- K6: reparameterisation + KL */
/* Model/sublayers.py:14-20 / Model/cvaetf.py:63-69: z = eps*exp(0.5*log_var)+mu.
 * eps_in nullable: then eps ~ N(0,1) is generated in-kernel (Philox + Box-Muller,
 * key (seed,site)) and written to eps_out (always written). */
int gct_reparam_fwd(const float* mu, const float* log_var, const float* eps_in, float* eps_out,
                    float* z, int64_t n, uint64_t seed, uint32_t site, void* stream);
/* dmu = dz + dmu_ext ; dlv = 0.5*dz*eps*exp(0.5*lv) + dlv_ext  (ext nullable) */
int gct_reparam_bwd(const float* dz, const float* log_var, const float* eps, const float* dmu_ext,
                    const float* dlv_ext, float* dmu, float* dlv, int64_t n, void* stream);
/* Train/trainer1.py:23  KLD = -0.5*sum(1+lv-mu^2-exp(lv)) over ALL n elements.
 * out[0] overwritten. ws >= 1024 floats. */
int gct_kld_fwd(const float* mu, const float* log_var, float* out, float* ws, int64_t n,
                void* stream);
/* dmu = g*mu ; dlv = g*0.5*(exp(lv)-1), g read from device scalar gout[0] */
int gct_kld_bwd(const float* mu, const float* log_var, const float* gout, float* dmu, float* dlv,
                int64_t n, void* stream);

/* ---------------------------------------------------------- K7: cross-entropy */
/* Train/trainer1.py:21-22  F.cross_entropy(logits.view(-1,V), ys, ignore_index=pad,
 * reduction='sum').  out[0] overwritten. ws >= 1024 floats. */
int gct_ce_fwd(const float* logits, const int64_t* target, float* out, float* ws, int64_t rows,
               int V, int64_t pad_id, void* stream);
/* dlogits = g*(softmax - onehot) on non-pad rows, 0 on pad rows; g = gout[0] (device). */
int gct_ce_bwd(const float* logits, const int64_t* target, const float* gout, float* dlogits,
               int64_t rows, int V, int64_t pad_id, void* stream);

/* -------------------------------------------------------------- K8: Adam step */
/* torch.optim.Adam semantics (train1.py:116-119; no weight decay, no amsgrad):
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * over one flat buffer of n elements; g is multiplied by gscale first (1/W folds the
 * data-parallel mean). `step` is t (1-based, already incremented). */
int gct_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                  float b2, float eps, int64_t step, float gscale, void* stream);
/* The same step behind a device-side guard: when *skip_if_nonzero != 0 at launch time nothing is updated (p, m, v stay
 * as they are).  The trainer passes the device counter of "gradient rows that fell on decoder rows the forward had
 * skipped" (gct_live_rows / gct_scatter_add_quads_check): a wrong gradient is then never applied, without a host
 * synchronisation in front of the update; the host raises at its next read-back.  skip_if_nonzero == NULL: no guard. */
int gct_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                          float b2, float eps, int64_t step, float gscale, const int32_t* skip_if_nonzero,
                          void* stream);

/* --------------------------------------------------------- K11: KV-cached decode */
/* Inference/sampling_tool.py:140-184 re-runs the whole decoder on ys[:, :i+1] each step; these
 * two kernels (with the GEMM/norm/embedding entry points above) make one step a fixed chain.
 * gct_attn_decode: ONE query row per (sample, head): q [n][H*dk] (ld ldq); keys/values row j of
 * sample b at k + b*kv_batch + j*kv_row (+ h*dk); valid (nullable) uint8 [n][>=Lc], 0 => -1e9;
 * o [n][H*dk].  Lc <= 256.
 * pos (nullable) = DEVICE-side step counter: the caches hold cache_off + *pos keys, and this step's own key / value
 * (row b of knew / vnew, leading dimension ldn) are appended at that row and attended to -- one captured graph then
 * serves every step of the loop (gct_decode_embed / gct_select_token read the same counter, gct_decode_advance
 * increments it at the end of the step).  With pos == NULL the first Lc cached keys are used as they are.
 * klen (nullable, only with pos == NULL): per-sample number of leading keys to look at -- for a key-padding mask
 * whose visible keys are a non-empty prefix, the masked rows behind it weigh exactly 0 and are not read. */
int gct_attn_decode(const float* q, int64_t ldq, float* k, float* v, int64_t kv_row,
                    int64_t kv_batch, const uint8_t* valid, int64_t valid_sb, float* o, int64_t ldo,
                    int n, int H, int Lc, int dk, float scale, const int32_t* pos, int cache_off,
                    const float* knew, const float* vnew, int64_t ldn, const int32_t* klen, void* stream);
/* gct_attn_decode_z: the cross-attention of a decode step (Model/layers.py:73-76 attn_2 over e_outputs = fc_z(z),
 * Model/vaetf.py:75-95) computed over the LATENT rows themselves.  k_{j,h} = G_h z_j + c_h and v_{j,h} = H_h z_j + d_h
 * with G_h = W_k,h W_z, H_h = W_v,h W_z, so score_{j,h} = (G_h^T q_h) . z_j + (a term equal for all keys of the row)
 * and sum_j p_j v_{j,h} = H_h sum_j p_j z_j + d_h: the caller folds G_h^T into the query projection and H_h into the
 * output projection once per sequence, and a step reads lat floats per key instead of 2 * d_model.
 *   q    [n][ldq]: columns [qoff, qoff + H*lat) hold q'_h = G_h^T q_h; columns [0, H*dk) the plain q (read only if nc > 0)
 *   z    row j of sample b at z + b*z_batch + j*lat, Le rows (lat % 4 == 0, lat <= 128, Le <= 256)
 *   ckv  (nc > 0: cond2lat memory, Model/cvaetf.py:86-96) the nc leading keys that are not functions of z: row j of
 *        sample b at ckv + b*ckv_batch + j*ld_ckv holds k'_j = k_j - c (H*dk floats) then v'_j = v_j - d (H*dk floats)
 *   valid (nullable) uint8 [n][>= nc + Le], 0 => -1e9 (condition rows first); klen (nullable) as in gct_attn_decode
 *   out  [n][ldo]: columns [ooff, ooff + H*lat) = sum_j p_j z_j per head; columns [0, H*dk) = sum_{j < nc} p_j v'_j
 *        (written only if nc > 0). */
int gct_attn_decode_z(const float* q, int64_t ldq, int qoff, const float* z, int64_t z_batch, int lat, int Le,
                      const float* ckv, int64_t ckv_batch, int64_t ld_ckv, int nc, const uint8_t* valid,
                      int64_t valid_sb, const int32_t* klen, float* out, int64_t ldo, int ooff, int n, int H, int dk,
                      float scale, void* stream);
/* x[b] = table[ys[b][*pos]] * scale + pe[pe_off + *pos] (Embeddings + PositionalEncoding of one position, eval mode) */
int gct_decode_embed(const int64_t* ys, int64_t ld_ys, const int32_t* pos, int pe_off, const float* table,
                     int vocab, const float* pe, float* out, int n, int d, float scale, void* stream);
int gct_decode_advance(int32_t* pos, void* stream);
/* softmax(logits[n][V]) then mode 0: argmax (first maximum, torch.max semantics) / mode 1:
 * multinomial (Philox inverse-CDF).  Writes ys[row*ld_ys + pos], valid[row*valid_sb + valid_off + pos] =
 * (token != pad), done[row] |= (token == eos); probs_out (nullable) [n][V].  pos_dev (nullable): pos = *pos_dev + 1;
 * seed_dev (nullable): the multinomial seed is read from device memory (graph replays with a fresh seed). */
int gct_select_token(const float* logits, int V, int64_t* ys, int64_t ld_ys, int pos, uint8_t* valid,
                     int64_t valid_sb, uint8_t* done, float* probs_out, int n, int mode,
                     int64_t pad_id, int64_t eos_id, uint64_t seed, const int32_t* pos_dev, int valid_off,
                     const uint64_t* seed_dev, void* stream);

/* ------------------------------------------------ host side: SMILES tokeniser + collate */
/* Utils/field.py:8-33 moltokenize (the atom-wise regex, findall semantics) as a scanner.
 * Returns the number of tokens (may exceed max_tokens: call again with a larger buffer). */
int gct_smiles_tokenize(const char* smi, int with_sep, int32_t* tok_start, int32_t* tok_len,
                        int max_tokens);
/* torchtext Field.process as used by Model/collate_fn.py:5-15,104-124: row r of out[n][width] =
 * [sos] ids(tokens) [eos] pad...; vocab[id] are the token strings, unknown -> unk_id; sos/eos < 0
 * omit them (SRC field).  Returns the longest row length, <0 on error (row does not fit). */
int gct_smiles_encode_batch(const char* const* smiles, int n, int with_sep, const char* const* vocab,
                            int vocab_size, int64_t unk_id, int64_t pad_id, int64_t sos_id,
                            int64_t eos_id, int64_t* out, int64_t width, int32_t* lengths);

/* ------------------------------------------------------------------ utilities */
/* dst[(r / rpb)*dst_rpb + dst_off + r % rpb][:] (op)= src[(r / rpb)*src_rpb + src_off + r % rpb][:]
 * row gather/scatter used for the cond2lat concat (Model/vaetf.py:88-91). accumulate: += */
int gct_copy_rows(const float* src, int64_t src_rpb, int64_t src_off, float* dst, int64_t dst_rpb,
                  int64_t dst_off, int64_t rows, int64_t rpb, int cols, int accumulate,
                  void* stream);
/* tiny dense layer for the property embeddings (Model/vaetf.py:30,75-77; K = n_c <= 8):
 * y[b][n] = sum_k x[b][k]*w[n][k] + b[n]; bwd gives dw, db (overwrite) and no dx. */
int gct_small_linear_fwd(const float* x, const float* w, const float* b, float* y, int rows,
                         int K, int N, void* stream);
int gct_small_linear_bwd(const float* dy, const float* x, float* dw, float* db, int rows, int K,
                         int N, void* stream);
/* dst[i] = sum_s slabs[s*stride + i] (deterministic slab reduction) */
int gct_reduce_slabs(const float* slabs, int nslab, int64_t stride, float* dst, int64_t n,
                     int accumulate, void* stream);
/* Deferred slab reductions: between gct_reduce_defer_begin and gct_reduce_defer_end every float4-shaped slab reduction
 * that the calling THREAD issues through this library (the tails of gct_linear_wgrad*, gct_norm_bwd's alpha / bias
 * partials, bias column sums) is recorded instead of launched; gct_reduce_defer_flush launches all recorded ones as ONE
 * kernel on `stream` (same summation order per region: results are bit-identical) and keeps recording, _end flushes and
 * stops, _pending returns the number recorded.  Contract: the caller keeps every recorded call's workspace intact until
 * the flush and does not read the destinations before it (a layer's parameter gradients: engine.py flushes at the end of
 * each layer's backward pass, before the data-parallel exchange may start). */
int gct_reduce_defer_begin(void);
int gct_reduce_defer_flush(void* stream);
int gct_reduce_defer_end(void* stream);
int gct_reduce_defer_pending(void);
/* y = a + b (gradient joins of the residual stream) */
int gct_add(const float* a, const float* b, float* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCTPLUS_HIP_H */
