// Host-side orchestration of the backward pass, one C call per reference module (the kernels themselves live in
// backward.hip / gemm.hip / wgrad_cu.hip).  The training step is host-bound in Python (~25 us of interpreter time per
// launch against ~15 us kernels on the dependency chain), so the launch sequences of a module's backward are issued
// from here, the way carca_forward issues the inference forward:
//   carca_sa_block_bwd      autograd of SelfAttentionBlock.forward   carca.py:297-318 (+ 228-265)
//   carca_cross_score_bwd   autograd of the final LayerNorm + CrossAttentionBlock.forward for every target group
//                                                                     carca.py:421, 338-349
//   carca_embed_bwd         autograd of AllEmbedding.forward over all row segments           carca.py:85-95
// Every entry launches its input-gradient chain on the caller's stream and APPENDS its weight-gradient products to a
// caller-owned host array; the caller launches them together (carca_gemm_wgrad_group) once the pass is issued --
// nothing downstream depends on them.
#include <string.h>

#include "carca_common.h"
#include "../../include/carca_hip.h"

namespace {

CarcaWgradDesc wgrad_product(const float* dy, int ld_dy, const float* x, int ld_x, int rows, int N, int K, float* dw,
                             int ldw, float* db) {
  CarcaWgradDesc w;
  memset(&w, 0, sizeof w);
  w.nseg = 1;
  w.seg[0].dy = dy;
  w.seg[0].x = x;
  w.seg[0].rows = rows;
  w.seg[0].T = 1;
  w.ld_dy = ld_dy;
  w.ld_x = ld_x;
  w.N = N;
  w.K = K;
  w.dw = dw;
  w.ldw = ldw;
  w.db = db;
  return w;
}

CarcaGemmDesc gemm_product(const float* a0, int lda0, const float* bt0, int ldb0, int K0, int N, float* c, int ldc,
                           int rows) {
  CarcaGemmDesc g;
  memset(&g, 0, sizeof g);
  g.nseg = 1;
  g.seg[0].a0 = a0;
  g.seg[0].c = c;
  g.seg[0].rows = rows;
  g.seg[0].T = 1;
  g.lda0 = lda0;
  g.K0 = K0;
  g.bt0 = bt0;
  g.ldb0 = ldb0;
  g.N = N;
  g.ldc = ldc;
  g.ncols_out = ldc;
  g.gate_slope = 0.01f;
  return g;
}

}  // namespace

extern "C" size_t carca_sa_block_bwd_workspace(int B, int L, int d, int H) {
  int dpi, dhp, dpo;
  if (carca_padded_dims(d, H, &dpi, &dhp, &dpo) != CARCA_OK) return 0;
  return (size_t)B * L * (6 * (size_t)dpi + 3 * (size_t)dpo);
}

extern "C" int carca_sa_block_bwd(const CarcaSaBwdDesc* D, CarcaWgradDesc* wgrads, int* n_wgrads, void* stream) {
  CARCA_CHECK_ARG(D && wgrads && n_wgrads && *n_wgrads >= 0, "sa_block_bwd: null descriptor / product array");
  CARCA_CHECK_ARG(D->B >= 1 && D->L >= 1 && D->d >= 1 && D->H >= 1 && D->d % D->H == 0 && D->drop_p >= 0.f &&
                      D->drop_p < 1.f,
                  "sa_block_bwd: bad dims");
  CARCA_CHECK_ARG(D->ids && D->dy && D->x_in && D->qn && D->qh && D->kh && D->vh && D->r && D->s2 && D->h1 &&
                      D->wq_t && D->wk_t && D->wv_t && D->w1_t && D->w2_t && D->ln1_w && D->ln2_w && D->g_w1 && D->g_b1 &&
                      D->g_w2 && D->g_b2 && D->g_wq && D->g_wk && D->g_wv && D->g_bq && D->g_bk && D->g_bv &&
                      D->g_ln1_w && D->g_ln1_b && D->g_ln2_w && D->g_ln2_b && D->workspace && D->dx,
                  "sa_block_bwd: null pointer");
  const bool drop = D->drop_p > 0.f;
  CARCA_CHECK_ARG(!drop || (D->m_attn && D->m_ffn2), "sa_block_bwd: dropout was on but the keep-masks are missing");
  int dpi, dhp, dpo;
  if (int rc = carca_padded_dims(D->d, D->H, &dpi, &dhp, &dpo)) return rc;
  const int rows = D->B * D->L, d = D->d;
  const float bscale = drop ? 1.0f / (1.0f - D->drop_p) : 1.0f;
  float* ws = D->workspace;
  float* dyf = ws;                          ws += (size_t)rows * dpi;  // dy through dropout2 (only when dropout was on)
  float* dh1pre = ws;                       ws += (size_t)rows * dpi;
  float* ds = ws;                           ws += (size_t)rows * dpi;
  float* dr = ws;                           ws += (size_t)rows * dpi;
  float* dqn = ws;                          ws += (size_t)rows * dpi;
  float* dx_kv = ws;                        ws += (size_t)rows * dpi;
  float* dqh = ws;                          ws += (size_t)rows * dpo;
  float* dkh = ws;                          ws += (size_t)rows * dpo;
  float* dvh = ws;
  int rc;
  // Without dropout the two row chains run as ONE launch each (row_chain.hip) around the attention core: 3 launches per
  // block instead of 6 (tuning key 6 = 1 or 2: the launch-per-step path below).
  if (!drop && dpi == dpo && dpi <= 96 && carca_tuning(6) == 0) {  // (dpi = 128: 16 waves leave 128 registers, the chain kernel would spill)
    if ((rc = carca_sa_ffn_chain_bwd(D->dy, D->h1, D->r, D->w2_t, D->w1_t, D->ln2_w, rows, d, dpi, D->residual, dh1pre, dr,
                                     D->g_ln2_w, D->g_ln2_b, stream)))
      return rc;
    if ((rc = carca_sa_attn_bwd(D->qh, D->kh, D->vh, dr, dpi, D->ids, dqh, dkh, dvh, D->B, D->L, d, D->H, nullptr, 1.0f,
                                stream)))
      return rc;
    if ((rc = carca_sa_input_chain_bwd(dqh, dkh, dvh, dr, D->x_in, D->wq_t, D->wk_t, D->wv_t, D->ln1_w, rows, d, dpi,
                                       D->residual, D->dx, D->g_ln1_w, D->g_ln1_b, stream)))
      return rc;
    CarcaWgradDesc* w = wgrads + *n_wgrads;
    w[0] = wgrad_product(D->dy, dpi, D->h1, dpi, rows, d, d, D->g_w2, d, D->g_b2);     // d ffn_2
    w[1] = wgrad_product(dh1pre, dpi, D->s2, dpi, rows, d, d, D->g_w1, d, D->g_b1);   // d ffn_1
    w[2] = wgrad_product(dqh, dpo, D->qn, dpi, rows, dpo, d, D->g_wq, d, D->g_bq);    // d W_Q (head-padded staging)
    w[3] = wgrad_product(dkh, dpo, D->x_in, dpi, rows, dpo, d, D->g_wk, d, D->g_bk);  // d W_K
    w[4] = wgrad_product(dvh, dpo, D->x_in, dpi, rows, dpo, d, D->g_wv, d, D->g_bv);  // d W_V
    *n_wgrads += 5;
    return CARCA_OK;
  }
  // f = dropout2(ffn_2(dropout1(lrelu(ffn_1(s))))) (+ s): the f branch sees dy * mask2 / (1 - p)   (carca.py:305-316)
  const float* dyf_c = D->dy;
  if (drop) {
    if ((rc = carca_mask_mul(D->dy, dpi, D->m_ffn2, dpi, bscale, dyf, dpi, rows, d, dpi, stream))) return rc;
    dyf_c = dyf;
  }
  {  // d h1pre = (dyf W_2) * LeakyReLU'(h1) (x dropout1')
    CarcaGemmDesc g = gemm_product(dyf_c, dpi, D->w2_t, dpi, d, d, dh1pre, dpi, rows);
    g.seg[0].gate = D->h1;
    g.ld_gate = dpi;
    g.gate_scale = bscale;
    g.gate_zero_drops = drop ? 1 : 0;
    if ((rc = carca_gemm_rows(&g, stream))) return rc;
  }
  {  // d s = d h1pre W_1 (+ dy: the post-LayerNorm2 residual)
    CarcaGemmDesc g = gemm_product(dh1pre, dpi, D->w1_t, dpi, d, d, ds, dpi, rows);
    if (D->residual) {
      g.seg[0].add = D->dy;
      g.ld_add = dpi;
    }
    if ((rc = carca_gemm_rows(&g, stream))) return rc;
  }
  // s = LayerNorm2(r), r = attention (+ q)
  if ((rc = carca_layernorm_bwd(ds, dpi, D->r, dpi, D->ln2_w, rows, d, nullptr, 0, dr, dpi, dpi, D->g_ln2_w, D->g_ln2_b,
                                stream)))
    return rc;
  if ((rc = carca_sa_attn_bwd(D->qh, D->kh, D->vh, dr, dpi, D->ids, dqh, dkh, dvh, D->B, D->L, d, D->H,
                              drop ? D->m_attn : nullptr, bscale, stream)))
    return rc;
  {  // d qn = dQ W_Q (+ dr: the normed residual) beside d x|kv = dK W_K + dV W_V: independent, one launch
    CarcaGemmDesc g[2];
    g[0] = gemm_product(dqh, dpo, D->wq_t, dpo, dpo, d, dqn, dpi, rows);
    if (D->residual) {
      g[0].seg[0].add = dr;
      g[0].ld_add = dpi;
    }
    g[1] = gemm_product(dkh, dpo, D->wk_t, dpo, dpo, d, dx_kv, dpi, rows);
    g[1].seg[0].a1 = dvh;
    g[1].lda1 = dpo;
    g[1].K1 = dpo;
    g[1].bt1 = D->wv_t;
    g[1].ldb1 = dpo;
    if ((rc = carca_gemm_rows_group(g, 2, stream))) return rc;
  }
  // q = LayerNorm1(x); K, V from x itself
  if ((rc = carca_layernorm_bwd(dqn, dpi, D->x_in, dpi, D->ln1_w, rows, d, dx_kv, dpi, D->dx, dpi, dpi, D->g_ln1_w,
                                D->g_ln1_b, stream)))
    return rc;
  CarcaWgradDesc* w = wgrads + *n_wgrads;
  w[0] = wgrad_product(dyf_c, dpi, D->h1, dpi, rows, d, d, D->g_w2, d, D->g_b2);    // d ffn_2
  w[1] = wgrad_product(dh1pre, dpi, D->s2, dpi, rows, d, d, D->g_w1, d, D->g_b1);   // d ffn_1
  w[2] = wgrad_product(dqh, dpo, D->qn, dpi, rows, dpo, d, D->g_wq, d, D->g_bq);    // d W_Q (head-padded staging)
  w[3] = wgrad_product(dkh, dpo, D->x_in, dpi, rows, dpo, d, D->g_wk, d, D->g_bk);  // d W_K
  w[4] = wgrad_product(dvh, dpo, D->x_in, dpi, rows, dpo, d, D->g_wv, d, D->g_bv);  // d W_V
  *n_wgrads += 5;
  return CARCA_OK;
}

extern "C" size_t carca_cross_score_bwd_workspace(int B, int L, int d, int H, const int32_t* Ns, int ngroups) {
  int dpi, dhp, dpo;
  if (carca_padded_dims(d, H, &dpi, &dhp, &dpo) != CARCA_OK || !Ns) return 0;
  size_t n = (size_t)B * L * (2 * (size_t)dpo + dpi);  // dK, dV, d p_normed
  for (int g = 0; g < ngroups; ++g) n += (size_t)B * Ns[g] * (dpo + 1) + 3;  // dQ_g, dlogit_g (+ alignment slack)
  return n;
}

extern "C" int carca_cross_score_bwd(const CarcaCrossBwdDesc* D, CarcaWgradDesc* wgrads, int* n_wgrads, void* stream) {
  CARCA_CHECK_ARG(D && wgrads && n_wgrads && *n_wgrads >= 0, "cross_score_bwd: null descriptor / product array");
  CARCA_CHECK_ARG(D->B >= 1 && D->L >= 1 && D->d >= 1 && D->H >= 1 && D->d % D->H == 0 && D->ngroups >= 1 &&
                      D->ngroups <= CARCA_MAX_GROUPS && D->drop_p >= 0.f && D->drop_p < 1.f,
                  "cross_score_bwd: bad dims");
  CARCA_CHECK_ARG(D->p_ids && D->kh && D->vh && D->p_normed && D->wq_t && D->wk_t && D->wv_t &&
                      D->ffn_w_pad && D->ffn_w && D->g_ffn_w && D->g_ffn_b && D->g_ffn_w_pad && D->g_wq &&
                      D->g_wk && D->g_wv && D->g_bq && D->g_bk && D->g_bv && D->workspace && D->dx,
                  "cross_score_bwd: null pointer");
  // norm_w = NULL: the stand-alone CrossAttentionBlock (carca.py:338-349 called on an already normed profile): no final
  // LayerNorm in front of it, dx = d p_normed
  const bool has_norm = D->norm_w != nullptr;
  CARCA_CHECK_ARG(!has_norm || (D->enc_out && D->g_norm_w && D->g_norm_b),
                  "cross_score_bwd: the final LayerNorm needs enc_out and its two gradient buffers");
  int dpi, dhp, dpo;
  if (int rc = carca_padded_dims(D->d, D->H, &dpi, &dhp, &dpo)) return rc;
  const int rows = D->B * D->L, d = D->d, ng = D->ngroups;
  const bool drop = D->drop_p > 0.f;
  float* ws = D->workspace;
  float* dkh = ws;  ws += (size_t)rows * dpo;
  float* dvh = ws;  ws += (size_t)rows * dpo;
  float* dp = ws;   ws += (size_t)rows * dpi;
  CarcaCrossBwdGroup grp[CARCA_MAX_GROUPS];
  float* dls[CARCA_MAX_GROUPS];
  for (int g = 0; g < ng; ++g) {
    const CarcaCrossBwdIn& in = D->group[g];
    CARCA_CHECK_ARG(in.qh && in.y && in.dy && in.ids && in.o && in.de && in.N >= 1 && (!drop || in.m_attn),
                    "cross_score_bwd: group %d malformed", g);
    const size_t gr = (size_t)D->B * in.N;
    grp[g].qh = in.qh; grp[g].y = in.y; grp[g].dy = in.dy; grp[g].ids = in.ids;
    grp[g].dqh = ws;                     ws += gr * dpo;
    grp[g].dlogit = dls[g] = ws;         ws += (gr + 3) / 4 * 4;
    grp[g].m_attn = drop ? in.m_attn : nullptr;
    grp[g].N = in.N;
    grp[g].ld_y = in.ld_y;
  }
  int rc;
  // attention core + sigmoid(ffn(.)) head: dQ per group, dK, dV, dlogit, the attention part of d ffn.weight
  if ((rc = carca_cross_attn_bwd(D->kh, D->vh, D->p_ids, grp, ng, D->ffn_w_pad, dkh, dvh, D->g_ffn_w_pad, D->B, D->L, d,
                                 D->H, D->training, drop ? 1.0f / (1.0f - D->drop_p) : 1.0f, stream)))
    return rc;
  {  // d o_g = dQ_g W_Q (+ dlogit (x) w), masked like e * mask (carca.py:94), beside d p_normed = dK W_K + dV W_V
    CarcaGemmDesc g[2];
    memset(g, 0, sizeof g);
    g[0].nseg = ng;
    for (int i = 0; i < ng; ++i) {
      CarcaGemmSeg& sg = g[0].seg[i];
      sg.a0 = grp[i].dqh; sg.c = D->group[i].de; sg.ids = grp[i].ids; sg.rows = D->B * grp[i].N; sg.T = 1;
      sg.rowscale = D->residual ? dls[i] : nullptr;
    }
    g[0].lda0 = dpo; g[0].K0 = dpo; g[0].bt0 = D->wq_t; g[0].ldb0 = dpo;
    // (inside CARCA the targets were embedded as e * mask: d e is masked the same way.  The stand-alone block's o is an
    // input of its own: a pad target still scores sigma(w . o + b) through the residual, and d o = dlogit (x) w there)
    g[0].N = d; g[0].ldc = dpi; g[0].ncols_out = dpi; g[0].gate_slope = 0.01f; g[0].mask_rows = has_norm ? 1 : 0;
    g[0].colvec = D->residual ? D->ffn_w : nullptr;
    g[1] = gemm_product(dkh, dpo, D->wk_t, dpo, dpo, d, has_norm ? dp : D->dx, dpi, rows);
    g[1].seg[0].a1 = dvh; g[1].lda1 = dpo; g[1].K1 = dpo; g[1].bt1 = D->wv_t; g[1].ldb1 = dpo;
    if ((rc = carca_gemm_rows_group(g, 2, stream))) return rc;
  }
  // final LayerNorm (carca.py:421)
  if (has_norm && (rc = carca_layernorm_bwd(dp, dpi, D->enc_out, dpi, D->norm_w, rows, d, nullptr, 0, D->dx, dpi, dpi,
                                            D->g_norm_w, D->g_norm_b, stream)))
    return rc;
  CarcaWgradDesc* w = wgrads + *n_wgrads;
  int n = 0;
  if (D->residual) {  // d ffn.bias = sum dlogit; residual part of d ffn.weight = dlogit^T o: a 1 x d product over the groups
    CarcaWgradDesc& p = w[n++];
    memset(&p, 0, sizeof p);
    p.nseg = ng;
    for (int i = 0; i < ng; ++i) {
      p.seg[i].dy = dls[i]; p.seg[i].x = D->group[i].o; p.seg[i].rows = D->B * grp[i].N; p.seg[i].T = 1;
    }
    p.ld_dy = 1; p.ld_x = dpi; p.N = 1; p.K = d; p.dw = D->g_ffn_w; p.ldw = d; p.db = D->g_ffn_b;
  } else {
    for (int i = 0; i < ng; ++i)
      if ((rc = carca_colsum(dls[i], 1, D->B * grp[i].N, 1, nullptr, nullptr, 1, D->g_ffn_b, stream))) return rc;
  }
  {  // d W_Q: one product over the groups' rows
    CarcaWgradDesc& p = w[n++];
    memset(&p, 0, sizeof p);
    p.nseg = ng;
    for (int i = 0; i < ng; ++i) {
      p.seg[i].dy = grp[i].dqh; p.seg[i].x = D->group[i].o; p.seg[i].rows = D->B * grp[i].N; p.seg[i].T = 1;
    }
    p.ld_dy = dpo; p.ld_x = dpi; p.N = dpo; p.K = d; p.dw = D->g_wq; p.ldw = d; p.db = D->g_bq;
  }
  w[n++] = wgrad_product(dkh, dpo, D->p_normed, dpi, rows, dpo, d, D->g_wk, d, D->g_bk);
  w[n++] = wgrad_product(dvh, dpo, D->p_normed, dpi, rows, dpo, d, D->g_wv, d, D->g_bv);
  *n_wgrads += n;
  return CARCA_OK;
}

extern "C" size_t carca_embed_bwd_workspace(const int32_t* rows, int nseg, int d, int g) {
  size_t n = 0;
  for (int s = 0; rows && s < nseg; ++s) n += (size_t)rows[s] * (d + g);  // d [z ; q] per segment
  return n;
}

namespace {
int g_early_recorded = 0;  // did the last carca_embed_bwd with an ev_early record it?  (process-wide: autograd runs the backward on a thread of its own)
}
extern "C" int carca_early_event_recorded(void) { return g_early_recorded; }

extern "C" int carca_embed_bwd(const CarcaEmbedBwdDesc* D, void* stream) {
  CARCA_CHECK_ARG(D && D->nseg >= 1 && D->nseg <= CARCA_MAX_SEGS, "embed_bwd: bad segment count");
  CARCA_CHECK_ARG(D->d >= 1 && D->g >= 1 && D->n_attrs >= 1 && D->n_ctx >= 0 && D->ld_de >= D->d && D->L >= 1,
                  "embed_bwd: bad dims");
  CARCA_CHECK_ARG(D->zq && D->joint_wt && D->g_items && D->g_feats_w && D->g_feats_b && D->g_joint_w && D->g_joint_b &&
                      (D->workspace || D->only_joint),  // (an only_joint call has no d [z ; q] to hold)
                  "embed_bwd: null pointer");
  const int d = D->d, g = D->g, ldz = d + g;
  int rc;
  CARCA_CHECK_ARG(!(D->only_joint && D->skip_joint), "embed_bwd: only_joint and skip_joint exclude each other");
  // d LearnableEncoding.encoding.weight[t] += sum over users of (d e * mask)[t]   (carca.py:25-31)
  if (D->g_pos) {
    const CarcaEmbedBwdSeg& p = D->seg[0];
    if ((rc = carca_colsum(p.de, D->ld_de, p.rows, d, nullptr, p.ids, D->L, D->g_pos, stream))) return rc;
  }
  CarcaWgradDesc wj, wf;
  CarcaGemmDesc gz;
  memset(&wj, 0, sizeof wj);
  memset(&wf, 0, sizeof wf);
  memset(&gz, 0, sizeof gz);
  size_t row0 = 0;
  float* ws = D->workspace;
  float* dzq[CARCA_MAX_SEGS];
  const int32_t* sids[CARCA_MAX_SEGS];
  int srows[CARCA_MAX_SEGS];
  int nj = 0, nf = 0;  // segments of d joint_embed / of the three products behind it
  for (int s = 0; s < D->nseg; ++s) {
    const CarcaEmbedBwdSeg& sg = D->seg[s];
    CARCA_CHECK_ARG(sg.rows >= 1 && sg.T >= 1 && sg.de && sg.ids, "embed_bwd: segment %d malformed", s);
    if (!D->skip_joint) {  // d joint_embed = (d e * mask)^T [z ; q]
      CarcaWgradSeg& j = wj.seg[nj++];
      j.dy = sg.de; j.x = D->zq + row0 * ldz; j.ids = sg.ids; j.rows = sg.rows; j.T = 1;
    }
    row0 += sg.rows;
    if (sg.joint_only || D->only_joint) continue;
    CARCA_CHECK_ARG((sg.attrs || sg.attrs_table) && (D->n_ctx == 0 || sg.ctx), "embed_bwd: segment %d malformed", s);
    dzq[nf] = ws;
    ws += (size_t)sg.rows * ldz;
    // d [z ; q] = (d e * mask) W_j
    gz.seg[nf].a0 = sg.de; gz.seg[nf].c = dzq[nf]; gz.seg[nf].ids = sg.ids; gz.seg[nf].rows = sg.rows; gz.seg[nf].T = 1;
    // d feats_embed = dq^T [attrs | ctx]: the ctx columns ride along as a second X source
    CarcaWgradSeg& f = wf.seg[nf];
    f.dy = dzq[nf] + d; f.rows = sg.rows; f.T = sg.T; f.ids = sg.ids;
    if (sg.attrs_table) {
      f.x = sg.attrs_table; f.x_gather = sg.attrs_table_rows > 1 ? sg.attrs_table_rows : 1;
    } else {
      f.x = sg.attrs; f.x_bstride = sg.attrs_bstride;
    }
    f.x1 = D->n_ctx ? sg.ctx : nullptr; f.x1_bstride = sg.ctx_bstride;
    sids[nf] = sg.ids; srows[nf] = sg.rows;
    ++nf;
  }
  CARCA_CHECK_ARG(nf >= 1 || D->only_joint, "embed_bwd: every segment is joint_only");
  wj.nseg = nj;
  gz.nseg = wf.nseg = nf;
  wf.ld_dy = ldz; wf.ld_x = D->n_attrs; wf.ld_x1 = D->n_ctx; wf.N = g; wf.K = D->n_attrs; wf.K1 = D->n_ctx;
  wf.dw = D->g_feats_w; wf.ldw = D->n_attrs + D->n_ctx; wf.db = D->g_feats_b;
  // (rows of pad items: their d q is zero already -- d [z ; q] below is masked -- but SAYING so lets the weight-gradient
  // kernel leave them out of its row table instead of multiplying zeros: 47 % of a C2 training batch's rows)
  wf.mask_rows = 1;
  // the d feats_embed kernel's row table on its own stream, beside everything this call launches in front of that kernel
  // -- only when that kernel WILL be launched (an unused fork is an empty branch of a capture)
  if (D->table_stream && !D->only_joint && nf >= 1 && carca_wgrad_cu_suited(&wf))
    if ((rc = carca_wgrad_table_fork((hipStream_t)stream, (hipStream_t)D->table_stream))) return rc;
  if (nj) {
    wj.ld_dy = D->ld_de; wj.ld_x = ldz; wj.N = d; wj.K = ldz; wj.dw = D->g_joint_w; wj.ldw = ldz; wj.db = D->g_joint_b;
    wj.mask_rows = 1;
    if ((rc = carca_gemm_wgrad(&wj, stream))) return rc;
  }
  if (D->only_joint) return CARCA_OK;
  gz.lda0 = D->ld_de; gz.K0 = d; gz.bt0 = D->joint_wt; gz.ldb0 = D->ld_joint_wt; gz.N = ldz; gz.ldc = ldz;
  gz.ncols_out = ldz; gz.gate_slope = 0.01f; gz.mask_rows = 1;
  if ((rc = carca_gemm_rows(&gz, stream))) return rc;
  // nn.Embedding(padding_idx = 0): z = E[ids] * sqrt(d); every segment in one launch
  if ((rc = carca_embed_scatter_segs(dzq, ldz, sids, srows, nf, d, (float)sqrt((double)d), D->g_items, stream))) return rc;
  if (D->ev_early) {
    // While the stream is being captured the record must become an EXTERNAL event-record node of the graph: every replay
    // then records the caller's event when it gets here, and a stream outside the graph -- the one that all-reduces the
    // early gradients -- can wait on it (a plain record inside a capture is only an edge of the capture).  Two ways to say
    // so; a runtime that takes neither leaves the event unrecorded and says so (carca_early_event_recorded): the caller
    // then reduces the early range behind the replay.
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t ev = (hipEvent_t)D->ev_early;
    g_early_recorded = 0;
    if (!carca_stream_capturing(st)) {
      if (hipEventRecord(ev, st) != hipSuccess) {
        carca_set_error("embed_bwd: cannot record the early-gradients event");
        return CARCA_ERR_BADARG;
      }
      g_early_recorded = 1;
    } else if (hipError_t e1 = hipEventRecordWithFlags(ev, st, hipEventRecordExternal); e1 == hipSuccess) {
      g_early_recorded = 1;
    } else {
      (void)hipGetLastError();
      hipError_t e2 = hipSuccess, e3 = hipSuccess, e4 = hipSuccess;
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      unsigned long long id = 0;
      hipGraph_t graph = nullptr;
      const hipGraphNode_t* deps = nullptr;
      size_t ndeps = 0;
      hipGraphNode_t node = nullptr;
      if ((e2 = hipStreamGetCaptureInfo_v2(st, &cs, &id, &graph, &deps, &ndeps)) == hipSuccess && cs == hipStreamCaptureStatusActive &&
          graph && (e3 = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, ev)) == hipSuccess &&
          (e4 = hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies)) == hipSuccess)
        g_early_recorded = 1;
      else {
        (void)hipGetLastError();
        carca_set_error("embed_bwd: no external event node in this capture (record-with-flags: %s; capture info: %s, graph %p, %zu deps; "
                        "add node: %s; update dependencies: %s)", hipGetErrorString(e1), hipGetErrorString(e2), (void*)graph, ndeps,
                        hipGetErrorString(e3), hipGetErrorString(e4));
      }
    }
  }
  rc = carca_gemm_wgrad(&wf, stream);
  const int rj = carca_wgrad_table_join((hipStream_t)stream);  // (a product that did not take the persistent kernel: close the fork)
  return rc ? rc : rj;
}
