// Epilogue shared by the row-GEMM kernels (gemm.hip, gemm_split.hip): include inside the translation unit, after
// carca_common.h and include/carca_hip.h.
#pragma once
namespace {
// The epilogue of the row GEMMs: D row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col = lane & 31 of each 32-column tile.
// Every operand it reads per row / per element is requested BEFORE the first is used -- the row ids and row scales of the
// lane's sixteen rows at once, then per column tile the sixteen addends / gates / position terms -- from clamped addresses
// inside wave-uniform branches: written as loads at their point of use (under `row < rows`, `n < N`, `if (ids)` ...) every
// one of them was a round trip of its own, 16 per tile (d [z;q], a K = 96 product with `mask_rows`: 42 us, "mostly epilogue").
template <int TN>
__device__ __forceinline__ void gemm_rows_epilogue(const CarcaGemmDesc& D, const CarcaGemmSeg& sg, const f32x16 (&acc)[TN],
                                                   const int n0, const int row_w, const int lr, const int lh) {
  int rid[16];
  float rsc[16];
  const int last = sg.rows - 1;
  if (D.mask_rows || D.add_table) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rid[r] = sg.ids[min(row_w + (r & 3) + 8 * (r >> 2) + 4 * lh, last)];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) rid[r] = 1;
  }
  if (sg.rowscale) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rsc[r] = sg.rowscale[min(row_w + (r & 3) + 8 * (r >> 2) + 4 * lh, last)];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) rsc[r] = 0.f;
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + tn * 32 + lr;
    if (n >= D.ncols_out) continue;
    const bool n_ok = n < D.N;
    const int nc = n_ok ? n : D.N - 1;
    const float bias = (n_ok && D.bias) ? D.bias[n] : 0.f;
    const float cv = (n_ok && D.colvec) ? D.colvec[n] : 0.f;
    float posv[16], addv[16], gatev[16], tabv[16];
    if (D.add_table) {  // (the addend gathered by id: CarcaGemmDesc.add_table)
#pragma unroll
      for (int r = 0; r < 16; ++r) tabv[r] = D.add_table[(size_t)rid[r] * D.ld_add_table + nc];
    }
    if (sg.add_pos) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        posv[r] = D.pos[(size_t)(min(row_w + (r & 3) + 8 * (r >> 2) + 4 * lh, last) % sg.T) * D.N + nc];
    }
    if (sg.add) {
#pragma unroll
      for (int r = 0; r < 16; ++r) addv[r] = sg.add[(size_t)min(row_w + (r & 3) + 8 * (r >> 2) + 4 * lh, last) * D.ld_add + nc];
    }
    if (sg.gate) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gatev[r] = sg.gate[(size_t)min(row_w + (r & 3) + 8 * (r >> 2) + 4 * lh, last) * D.ld_gate + nc];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_w + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= sg.rows) continue;
      float v = 0.f;
      if (n_ok) {
        v = (D.alpha != 0.f ? D.alpha * acc[tn][r] : acc[tn][r]) + bias;
        if (sg.add_pos) v += posv[r];
        if (sg.add) v += addv[r];
        if (D.add_table) v += tabv[r];
        if (sg.rowscale) v += rsc[r] * cv;
        if (sg.gate) {
          const float gv = gatev[r];
          const float gs = D.gate_scale != 0.f ? D.gate_scale : 1.0f;
          v *= gv > 0.f ? gs : ((gv < 0.f || !D.gate_zero_drops) ? D.gate_slope * gs : 0.f);
        }
        if (D.mask_rows) v = rid[r] != 0 ? v : 0.f;  // e * mask (carca.py:94): exact zeros
      }
      sg.c[(size_t)row * D.ldc + n] = v;
    }
  }
}

}  // namespace
