// Host-side exerciser of libmi355pose's entry points for the CPU-box sanitizer job (tests/test_host_sanitizers.py): the HOST pass
// of every .hip file is compiled with -fsanitize=address,undefined (hipcc --cuda-host-only) and linked with this program.
// No GPU is needed or used: device pointers are never dereferenced on the host (fake, well-aligned addresses are passed), every
// kernel launch fails in the HIP runtime ("no device") AFTER the host code under test has run -- descriptor checks, tap / phase
// tables of the strided input gradient, split-count plans, the grouped weight-gradient argument blocks, workspace sizing, the
// concat-K / weights-stationary GEMM dispatch arithmetic.  The job passes when no sanitizer report aborts the process and every
// invalid call is refused with its error code.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/mi355pose.h"

static int g_fail = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d  %s  (last error: %s)\n", __FILE__, __LINE__, #cond, mi355_last_error()); ++g_fail; } } while (0)
static void* fake(size_t i) { return reinterpret_cast<void*>(static_cast<uintptr_t>(0x100000000ull + (i << 28))); }

static mi355_conv_desc desc(int N, int H, int W, int Ci, int Co, int k, int s, int p, int dtype) {
  mi355_conv_desc d; std::memset(&d, 0, sizeof(d));
  d.N = N; d.Hi = H; d.Wi = W; d.Ci = Ci; d.Co = Co; d.kh = d.kw = k; d.stride = s; d.pad = p; d.dtype = dtype;
  d.Ho = (H + 2 * p - k) / s + 1; d.Wo = (W + 2 * p - k) / s + 1;
  return d;
}

int main() {
  EXPECT(mi355_version() >= 100);
  // ---- every conv geometry of the ResNet-50 iteration (SURVEY table A2) through forward, input gradient (phase / tap tables for
  // stride 2), weight gradient (all three kernels' plans), with and without the statistics epilogue; launches fail (no GPU), the
  // host code in front of them must be clean
  struct G { int N, H, Ci, Co, k, s, p; };
  const G geo[] = {{64, 256, 8, 64, 7, 2, 3}, {64, 64, 64, 64, 1, 1, 0}, {64, 64, 64, 64, 3, 1, 1}, {64, 64, 64, 256, 1, 1, 0}, {64, 64, 256, 64, 1, 1, 0},
                   {64, 64, 256, 128, 1, 1, 0}, {64, 64, 128, 128, 3, 2, 1}, {64, 32, 128, 512, 1, 1, 0}, {64, 64, 256, 512, 1, 2, 0},
                   {64, 32, 512, 128, 1, 1, 0}, {64, 32, 128, 128, 3, 1, 1}, {64, 32, 512, 256, 1, 1, 0}, {64, 32, 256, 256, 3, 2, 1},
                   {64, 16, 256, 1024, 1, 1, 0}, {64, 32, 512, 1024, 1, 2, 0}, {64, 16, 1024, 256, 1, 1, 0}, {64, 16, 256, 256, 3, 1, 1},
                   {64, 16, 1024, 512, 1, 1, 0}, {64, 16, 512, 512, 3, 2, 1}, {64, 8, 512, 2048, 1, 1, 0}, {64, 16, 1024, 2048, 1, 2, 0},
                   {64, 8, 2048, 512, 1, 1, 0}, {64, 8, 512, 512, 3, 1, 1}, {64, 16, 256, 2048, 4, 2, 1}, {64, 32, 256, 256, 4, 2, 1},
                   {64, 64, 256, 256, 4, 2, 1}, {64, 64, 256, 256, 3, 1, 1}, {64, 64, 256, 256, 3, 2, 1}, {64, 64, 256, 256, 1, 1, 0},
                   {3, 9, 64, 64, 3, 1, 1}, {5, 12, 128, 64, 3, 2, 1}, {1, 13, 64, 128, 3, 2, 1}, {2, 7, 64, 64, 1, 1, 0}};
  std::vector<float> partial_host(4);     // (never written: the launch fails first) -- only its size matters
  for (int dt = MI355_F32; dt <= MI355_BF16; ++dt) {
    for (const G& g : geo) {
      if (dt == MI355_F32 && g.Ci % 4) continue;
      mi355_conv_desc d = desc(g.N, g.H, g.H, g.Ci, g.Co, g.k, g.s, g.p, dt);
      int ns = -1;
      const size_t sb = mi355_conv_stats_bytes((long)d.N * d.Ho * d.Wo, d.Co);
      EXPECT(sb > 0);
      int rc = mi355_conv_fwd(&d, fake(1), fake(2), nullptr, nullptr, fake(3), nullptr);
      EXPECT(rc == MI355_ELAUNCH || rc == MI355_OK);
      rc = mi355_conv_fwd_stats(&d, fake(1), fake(2), (const float*)fake(4), fake(3), (float*)fake(5), sb, &ns, nullptr);
      EXPECT(rc == MI355_ELAUNCH || rc == MI355_OK);
      rc = mi355_conv_dgrad(&d, fake(3), fake(2), nullptr, (const float*)fake(6), 0, fake(1), nullptr);
      EXPECT(rc == MI355_ELAUNCH || rc == MI355_OK);
      rc = mi355_conv_dgrad(&d, fake(3), fake(2), nullptr, nullptr, 1, fake(1), nullptr);
      EXPECT(rc == MI355_ELAUNCH || rc == MI355_OK);
      const size_t sbd = mi355_conv_stats_bytes((long)d.N * d.Hi * d.Wi, d.Ci);
      rc = mi355_conv_dgrad_stats(&d, fake(3), fake(2), fake(1), (float*)fake(5), sbd, &ns, nullptr);
      EXPECT(rc == MI355_ELAUNCH || rc == MI355_OK);
      const size_t ws = mi355_conv_wgrad_workspace(&d);
      rc = mi355_conv_wgrad(&d, fake(1), fake(3), (float*)fake(7), 0, fake(8), ws, nullptr);
      EXPECT(rc == MI355_ELAUNCH || rc == MI355_OK);
      rc = mi355_conv_wgrad(&d, fake(1), fake(3), (float*)fake(7), 1, fake(8), ws ? ws - 1 : 0, nullptr);     // one byte short, accumulate
      EXPECT(rc == MI355_EWORKSPACE || rc == MI355_ELAUNCH);
      // concat-K forward (second operand 32 channels at the output resolution)
      if (g.Ci >= 64) {
        rc = mi355_conv_fwd_cat(&d, fake(1), fake(2), nullptr, fake(9), fake(10), nullptr, 32, fake(3), (float*)fake(5), sb, &ns, nullptr);
        EXPECT(rc == MI355_ELAUNCH || rc == MI355_OK);
        rc = mi355_conv_fwd_cat(&d, fake(1), fake(2), nullptr, fake(9), fake(10), nullptr, 128, fake(3), nullptr, 0, nullptr, nullptr);
        EXPECT(rc == MI355_EINVAL);
      }
    }
  }
  // ---- grouped weight gradients: one ResNet stage's small problems, an item repeated (shared dw), a mixed-dtype list, the cap of 24
  {
    std::vector<mi355_wgrad_item> items;
    auto add = [&](const G& g, int dt, size_t dw, int acc) {
      mi355_wgrad_item it; std::memset(&it, 0, sizeof(it));
      it.d = desc(g.N, g.H, g.H, g.Ci, g.Co, g.k, g.s, g.p, dt); it.x = fake(1); it.dy = fake(3); it.dw = (float*)fake(20 + dw); it.accumulate = acc;
      items.push_back(it);
    };
    for (int r = 0; r < 6; ++r) { add({64, 16, 256, 1024, 1, 1, 0}, MI355_BF16, 2 * r, 0); add({64, 16, 1024, 256, 1, 1, 0}, MI355_BF16, 2 * r + 1, 0); add({64, 16, 256, 256, 3, 1, 1}, MI355_BF16, 40 + r, 0); }
    add({64, 16, 256, 1024, 1, 1, 0}, MI355_BF16, 0, 1);                 // same dw as the first item: must close the pending group
    add({64, 32, 512, 1024, 1, 2, 0}, MI355_BF16, 60, 0);
    add({2, 7, 64, 64, 1, 1, 0}, MI355_F32, 61, 0);                      // other dtype: its own group
    for (int r = 0; r < 30; ++r) add({64, 8, 512, 2048, 1, 1, 0}, MI355_BF16, 70 + r, 0);      // more than one launch's worth
    const size_t ws = mi355_conv_wgrad_grouped_workspace(items.data(), (int)items.size());
    EXPECT(ws > 0);
    int rc = mi355_conv_wgrad_grouped(items.data(), (int)items.size(), fake(8), ws, nullptr);
    EXPECT(rc == MI355_ELAUNCH || rc == MI355_OK);
    rc = mi355_conv_wgrad_grouped(items.data(), (int)items.size(), fake(8), 16, nullptr);
    EXPECT(rc == MI355_EWORKSPACE || rc == MI355_ELAUNCH);
    items[0].x = nullptr;                                            // (item 0: the check runs before any launch is attempted)
    EXPECT(mi355_conv_wgrad_grouped(items.data(), (int)items.size(), fake(8), ws, nullptr) == MI355_EINVAL);
    EXPECT(mi355_conv_wgrad_grouped(nullptr, 0, nullptr, 0, nullptr) == MI355_EINVAL);
  }
  // ---- refused descriptors
  {
    mi355_conv_desc d = desc(2, 16, 16, 64, 64, 3, 1, 1, MI355_BF16);
    mi355_conv_desc b = d; b.Ho += 1;
    EXPECT(mi355_conv_fwd(&b, fake(1), fake(2), nullptr, nullptr, fake(3), nullptr) == MI355_EINVAL);
    // cropped outputs (top-left Ho x Wo of a unit-stride conv): forward and weight gradient take them, the input gradient and every
    // strided / 1x1 / fp8 descriptor do not; the folded stem's descriptor (4x4 over the 128 x 128 x 16 space-to-depth image) is one
    b = d; b.Ho -= 3; b.Wo -= 1;
    int rcc = mi355_conv_fwd(&b, fake(1), fake(2), nullptr, nullptr, fake(3), nullptr);
    EXPECT(rcc == MI355_ELAUNCH || rcc == MI355_OK);
    rcc = mi355_conv_wgrad(&b, fake(1), fake(3), (float*)fake(7), 0, fake(8), mi355_conv_wgrad_workspace(&b), nullptr);
    EXPECT(rcc == MI355_ELAUNCH || rcc == MI355_OK);
    EXPECT(mi355_conv_dgrad(&b, fake(3), fake(2), nullptr, nullptr, 0, fake(1), nullptr) == MI355_EINVAL);
    b = desc(2, 16, 16, 64, 64, 3, 2, 1, MI355_BF16); b.Ho -= 1;
    EXPECT(mi355_conv_fwd(&b, fake(1), fake(2), nullptr, nullptr, fake(3), nullptr) == MI355_EINVAL);
    b = desc(2, 16, 16, 64, 64, 1, 1, 0, MI355_BF16); b.Ho -= 1;
    EXPECT(mi355_conv_fwd(&b, fake(1), fake(2), nullptr, nullptr, fake(3), nullptr) == MI355_EINVAL);
    for (int dt = MI355_F32; dt <= MI355_BF16; ++dt) {
      b = desc(64, 128, 128, 16, 64, 4, 1, 2, dt); b.Ho = b.Wo = 128;
      int ns = -1;
      rcc = mi355_conv_fwd_stats(&b, fake(1), fake(2), nullptr, fake(3), (float*)fake(5), mi355_conv_stats_bytes(64L * 128 * 128, 64), &ns, nullptr);
      EXPECT(rcc == MI355_ELAUNCH || rcc == MI355_OK);
      rcc = mi355_conv_wgrad(&b, fake(1), fake(3), (float*)fake(7), 0, fake(8), mi355_conv_wgrad_workspace(&b), nullptr);
      EXPECT(rcc == MI355_ELAUNCH || rcc == MI355_OK);
      rcc = mi355_nchw_to_s2d((const float*)fake(1), fake(2), 64, 256, 256, dt, nullptr);
      EXPECT(rcc == MI355_ELAUNCH || rcc == MI355_OK);
      rcc = mi355_stem_s2d_pack((const float*)fake(1), fake(2), 64, dt, nullptr);
      EXPECT(rcc == MI355_ELAUNCH || rcc == MI355_OK);
    }
    rcc = mi355_stem_s2d_unpack_grad((const float*)fake(1), (float*)fake(2), 64, 1, nullptr);
    EXPECT(rcc == MI355_ELAUNCH || rcc == MI355_OK);
    EXPECT(mi355_nchw_to_s2d((const float*)fake(1), fake(2), 2, 255, 256, MI355_BF16, nullptr) == MI355_EINVAL);      // odd extent
    EXPECT(mi355_nchw_to_s2d((const float*)fake(1), fake(2), 2, 256, 256, MI355_FP8, nullptr) == MI355_EINVAL);
    EXPECT(mi355_stem_s2d_pack(nullptr, fake(2), 64, MI355_BF16, nullptr) == MI355_EINVAL);
    EXPECT(mi355_stem_s2d_unpack_grad((const float*)fake(1), nullptr, 64, 0, nullptr) == MI355_EINVAL);
    b = d; b.stride = 3;
    EXPECT(mi355_conv_fwd(&b, fake(1), fake(2), nullptr, nullptr, fake(3), nullptr) == MI355_EINVAL);
    b = d; b.kh = b.kw = 9;
    EXPECT(mi355_conv_fwd(&b, fake(1), fake(2), nullptr, nullptr, fake(3), nullptr) == MI355_EINVAL);
    b = d; b.dtype = 17;
    EXPECT(mi355_conv_fwd(&b, fake(1), fake(2), nullptr, nullptr, fake(3), nullptr) == MI355_EINVAL);
    b = d; b.Ci = 48;                                                    // not a power-of-two chunk count
    EXPECT(mi355_conv_fwd(&b, fake(1), fake(2), nullptr, nullptr, fake(3), nullptr) == MI355_EINVAL);
    b = desc(4096, 512, 512, 64, 64, 3, 1, 1, MI355_BF16);               // > 2^31 bytes
    EXPECT(mi355_conv_fwd(&b, fake(1), fake(2), nullptr, nullptr, fake(3), nullptr) == MI355_EINVAL);
    EXPECT(mi355_conv_fwd(nullptr, fake(1), fake(2), nullptr, nullptr, fake(3), nullptr) == MI355_EINVAL);
    EXPECT(mi355_conv_dgrad_masked_acc(&d, fake(3), fake(2), nullptr, fake(1), nullptr, nullptr) == MI355_EINVAL);
    EXPECT(mi355_conv1x1_heatmap(fake(1), fake(2), nullptr, (float*)fake(3), 2, 64, 256, 64, MI355_BF16, nullptr) == MI355_EINVAL);   // K > 32
  }
  // ---- BatchNorm workspace sizing / argument checks
  {
    const long rows[] = {64L * 64 * 64, 64L * 8 * 8, 2L * 5 * 7, 64L * 128 * 128};
    const int chans[] = {64, 2048, 24, 256};
    for (long r : rows) for (int c : chans) EXPECT(mi355_bn_workspace(r, c) >= (size_t)(4 * c) * sizeof(float));
    EXPECT(mi355_colsum_workspace(4096, 256) > 0);
    int rc = mi355_bn_train_fwd(fake(1), nullptr, fake(2), (const float*)fake(3), (const float*)fake(4), nullptr, nullptr, nullptr, (float*)fake(5),
                                (float*)fake(6), 4096, 30, 1e-5f, 0.1f, 1, 1, MI355_BF16, fake(8), 1 << 20, nullptr, nullptr, nullptr, nullptr);
    EXPECT(rc == MI355_EINVAL);                                          // C not a multiple of the chunk
    rc = mi355_bn_train_fwd(fake(1), nullptr, fake(2), (const float*)fake(3), (const float*)fake(4), nullptr, nullptr, nullptr, (float*)fake(5),
                            (float*)fake(6), 4096, 64, 1e-5f, 0.1f, 1, 1, MI355_BF16, fake(8), 16, nullptr, nullptr, nullptr, nullptr);
    EXPECT(rc == MI355_EWORKSPACE);
    rc = mi355_bn_train_fwd(fake(1), nullptr, fake(2), (const float*)fake(3), (const float*)fake(4), nullptr, nullptr, nullptr, (float*)fake(5),
                            (float*)fake(6), 4096, 64, 1e-5f, 0.1f, 1, 1, MI355_BF16, fake(8), mi355_bn_workspace(4096, 64), nullptr, nullptr, nullptr, nullptr);
    EXPECT(rc == MI355_ELAUNCH || rc == MI355_OK);
  }
  std::printf("host sanitizer driver: %d failure(s)\n", g_fail);
  return g_fail ? 1 : 0;
}
