// Diagnostic: s_memtime stamps of attn_fwd_kernel (B=512,H=8,L=80,dk=64, key-padding mask, ragged lengths).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGCT_STAMPS tools/attn_stamps.hip -o tools/_build/attn_stamps
#include "../gct_plus_amd/csrc/capi.hip"
#include "../gct_plus_amd/csrc/attention.hip"
#include <vector>
int main() {
  const int B = 512, H = 8, L = 80, dk = 64, d = H * dk;
  float *qkv, *o, *lse; uint8_t* mask; uint32_t *bits, *tbits; unsigned long long* st;
  hipMalloc(&qkv, (size_t)B * L * 3 * d * 4); hipMalloc(&o, (size_t)B * L * d * 4); hipMalloc(&lse, B * H * L * 4);
  hipMalloc(&mask, B * L); hipMalloc(&bits, B * 8 * 4); hipMalloc(&tbits, B * 4); hipMalloc(&st, 64 * 6 * 8 * 8);
  std::vector<float> h((size_t)B * L * 3 * d);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int fixed = 0; fixed < 2; ++fixed) {
    std::vector<uint8_t> m(B * L);
    for (int b = 0; b < B; ++b) for (int j = 0; j < L; ++j) m[b * L + j] = fixed ? 1 : (j < 20 + (b * 7) % 40);
    hipMemcpy(mask, m.data(), m.size(), hipMemcpyHostToDevice);
    gct_attn_mask_pack(mask, L, 0, B, L, L, bits, tbits, nullptr);
    for (int variant = 0; variant < 2; ++variant) {
      AttnArgs a = {};
      a.q = qkv; a.k = qkv + d; a.v = qkv + 2 * d; a.ldq = a.ldk = a.ldv = 3 * d;
      a.mbits = bits; a.mb_sb = 8; a.mb_sq = 0; a.o = o; a.ldo = d; a.lse = lse;
      a.B = B; a.H = H; a.Lq = L; a.Lk = L; a.npairs = B * H; a.scale = 0.125f;
      const float p = variant ? 0.1f : 0.0f;
      a.thr = gct_drop_threshold(p); a.keep_scale = 1.f / (1.f - p); a.rng = gct_rng_make(1, 1);
      a.stamps = st;
      const size_t lds = (size_t)(160) * 68 * 4;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(st, 0, 64 * 6 * 8 * 8);
        hipLaunchKernelGGL((attn_fwd_kernel<4, 6, true, 3>), dim3(512), dim3(ATT_THREADS), lds, 0, a);
        hipDeviceSynchronize();
      }
      std::vector<unsigned long long> hs(64 * 6 * 8);
      hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
      const char* nm[6] = {"seam", "wait+lds store", "barrier1", "prefetch issue", "compute", "barrier2"};
      printf("fixed=%d dropout p=%.1f (cycles per wave over 8 pairs)\n", fixed, p);
      for (int w = 0; w < 6; ++w) {
        double seg[8] = {0};
        for (int b = 0; b < 64; ++b) for (int i = 0; i < 8; ++i) seg[i] += (double)hs[(b * 6 + w) * 8 + i] / 64;
        double tot = 0; for (int i = 0; i < 6; ++i) tot += seg[i];
        printf(" wave %d total %.0f:", w, tot);
        for (int i = 0; i < 6; ++i) printf("  %s %.0f", nm[i], seg[i]);
        printf("\n");
      }
    }
  }
  // ---- backward (dropout 0.1, ragged key-padding mask): same stamps, 8 segments
  {
    float *dout, *dqkv;
    hipMalloc(&dout, (size_t)B * L * d * 4); hipMalloc(&dqkv, (size_t)B * L * 3 * d * 4);
    hipMemcpy(dout, h.data(), (size_t)B * L * d * 4, hipMemcpyHostToDevice);
    std::vector<uint8_t> m(B * L);
    for (int b = 0; b < B; ++b) for (int j = 0; j < L; ++j) m[b * L + j] = (j < 20 + (b * 7) % 40);
    hipMemcpy(mask, m.data(), m.size(), hipMemcpyHostToDevice);
    gct_attn_mask_pack(mask, L, 0, B, L, L, bits, tbits, nullptr);
    AttnArgs a = {};
    a.q = qkv; a.k = qkv + d; a.v = qkv + 2 * d; a.ldq = a.ldk = a.ldv = 3 * d;
    a.mbits = bits; a.mb_sb = 8; a.mb_sq = 0; a.o_in = o; a.dout = dout; a.ldo = d; a.lse_in = lse;
    a.dq = dqkv; a.dk = dqkv + d; a.dv = dqkv + 2 * d; a.lddq = a.lddk = a.lddv = 3 * d;
    a.B = B; a.H = H; a.Lq = L; a.Lk = L; a.npairs = B * H; a.scale = 0.125f;
    a.thr = gct_drop_threshold(0.1f); a.keep_scale = 1.f / 0.9f; a.rng = gct_rng_make(1, 1);
    a.stamps = st;
    const size_t lds = (size_t)(160) * 68 * 4 + 80 * 8 + 80 * 3 * 8 + 160;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(st, 0, 64 * 6 * 8 * 8);
      hipLaunchKernelGGL((attn_bwd_kernel<4, 6>), dim3(1024), dim3(ATT_THREADS), lds, 0, a);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> hs(64 * 6 * 8);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    const char* nm[8] = {"seam", "loads+mask+lds store", "barrier1", "phase A", "own K/V + barrier2", "frag->lds", "barrier3", "phase B"};
    printf("backward, dropout 0.1, ragged (ticks per wave over 4 pairs)\n");
    for (int w = 0; w < 6; ++w) {
      double seg[8] = {0};
      for (int b = 0; b < 64; ++b) for (int i = 0; i < 8; ++i) seg[i] += (double)hs[(b * 6 + w) * 8 + i] / 64;
      double tot = 0; for (int i = 0; i < 8; ++i) tot += seg[i];
      printf(" wave %d total %.0f:", w, tot);
      for (int i = 0; i < 8; ++i) printf("  %s %.0f", nm[i], seg[i]);
      printf("\n");
    }
  }
  // ---- direct kernels (one wave per item): mean cycles per segment over items 8192 .. 12287, ragged key-padding
  //      mask, dropout 0.1; through the library entry points (g_attn_stamps is copied into the kernel arguments)
  {
    float *dout, *dqkv; void* ws; unsigned long long* st2;
    hipMalloc(&dout, (size_t)B * L * d * 4); hipMalloc(&dqkv, (size_t)B * L * 3 * d * 4);
    hipMemcpy(dout, h.data(), (size_t)B * L * d * 4, hipMemcpyHostToDevice);
    const int64_t wsb = gct_attn_bwd_ws_bytes(B, H, L, L);
    hipMalloc(&ws, wsb); hipMalloc(&st2, 2 * 4096 * 8 * 8);
    std::vector<uint8_t> m(B * L);
    for (int b = 0; b < B; ++b) for (int j = 0; j < L; ++j) m[b * L + j] = (j < 20 + (b * 7) % 40);
    hipMemcpy(mask, m.data(), m.size(), hipMemcpyHostToDevice);
    gct_attn_mask_pack(mask, L, 0, B, L, L, bits, tbits, nullptr);
    auto report = [&](const char* what, const char* const* nm, int nseg, int rec0 = 0) {
      std::vector<unsigned long long> hs(4096 * 8);
      hipMemcpy(hs.data(), st2 + (size_t)rec0 * 8, hs.size() * 8, hipMemcpyDeviceToHost);
      double seg[8] = {0}; int cnt = 0;
      for (int it = 0; it < 4096; ++it) {
        unsigned long long tot = 0;
        for (int i = 0; i < 8; ++i) tot += hs[it * 8 + i];
        if (!tot) continue;
        ++cnt;
        for (int i = 0; i < 8; ++i) seg[i] += (double)hs[it * 8 + i];
      }
      double tot = 0; for (int i = 0; i < nseg; ++i) tot += seg[i] / cnt;
      printf("%s: %d items, %.0f ticks per item:", what, cnt, tot);
      for (int i = 0; i < nseg; ++i) printf("  [%s] %.0f", nm[i], seg[i] / cnt);
      printf("\n");
    };
    g_attn_stamps = st2;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(st2, 0, 2 * 4096 * 8 * 8);
      gct_attn_fwd(qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, bits, 8, 0, o, d, lse, nullptr, B, H, L, L, dk, 0.125f, 0.1f, 1, 1,
                   nullptr, nullptr, tbits, 1, 0, nullptr);
      hipDeviceSynchronize();
    }
    const char* nf[7] = {"Q/mask req, mask, tiles", "K req, Q->frag", "K->frag, S issue", "V req", "S results, softmax, bits", "V arrival, PV issue", "PV results, store"};
    report("attn_fwd_direct", nf, 7);
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(st2, 0, 2 * 4096 * 8 * 8);
      g_attn_stamps = st2;
      gct_attn_bwd(qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, bits, 8, 0, o, dout, d, lse, dqkv, 3 * d, dqkv + d, 3 * d,
                   dqkv + 2 * d, 3 * d, B, H, L, L, dk, 0.125f, 0.1f, 1, 1, nullptr, nullptr, 0, nullptr, nullptr, tbits, 1, 0, ws, wsb, nullptr);
      hipDeviceSynchronize();
    }
    const char* nq[7] = {"Q/dO/O->frag, delta", "mask, scalars, tiles, bits", "K/V req (sum over t)", "K/V->frag, S/dP issue", "S/dP results, dS", "dQ issue", "results, store"};
    report("attn_bwd_dq (launch 1)", nq, 7, 4096);
    const char* nk[6] = {"visit list, K/V->frag", "Q/dO req (sum over u)", "Q/dO->frag, S/dP issue", "scalars, S/dP results, P/dS", "dV/dK issue", "results, store"};
    report("attn_bwd_dkv (launch 2)", nk, 6);
  }
  return 0;
}
