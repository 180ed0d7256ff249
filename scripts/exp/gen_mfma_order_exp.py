#!/usr/bin/env python3
"""Generates scripts/exp/native/mfma_order_power.hip: register-only MFMA loops whose instruction ORDER is written out (one asm
block per variant, explicit registers, accumulators in place), to price what the order of a K tile's MFMAs costs in energy.
The card sits on its 1400 W cap under the burner, so the sustained rate of a loop IS its energy per FLOP.

Every variant issues the same 128 MFMAs of 16x16x32 bf16 (or 32 of 32x32x16) per loop pass on the same 8 + 8 random fragments;
the second half of a pass runs on -A so the accumulators return to ~0 (a GEMM's accumulators are zero-mean random walks)."""
import os

A0, B0, AN = 40, 72, 104


def frag(base, i):
    return f"v[{base + 4 * i}:{base + 4 * i + 3}]"


def order16(kind):
    out = []
    for i in range(64):
        if kind == "m":
            m, n = i >> 3, i & 7
        elif kind == "m_snake":
            m = i >> 3
            n = 7 - (i & 7) if m & 1 else i & 7
        elif kind == "n":
            n, m = i >> 3, i & 7
        elif kind == "n_snake":
            n = i >> 3
            m = 7 - (i & 7) if n & 1 else i & 7
        elif kind == "diag":
            m, n = i & 7, ((i & 7) + (i >> 3)) & 7
        elif kind == "block2":      # 2 x 2 blocks of tiles, walked so that every step changes ONE source
            blk, j = i >> 2, i & 3
            bm, bn = blk >> 2, blk & 3
            if bm & 1:
                bn = 3 - bn
            dm, dn = ((0, 0), (0, 1), (1, 1), (1, 0))[j]
            m, n = 2 * bm + dm, 2 * bn + dn
        elif kind in ("zigzag2", "zigzag2n"):
            # two rows at a time, every step changes exactly ONE source: (r0,n) (r1,n) (r1,n') (r0,n') (r0,n'') ..
            p, j = i >> 4, i & 15
            col, k = j >> 1, j & 1
            n = 7 - col if p & 1 else col
            r = (k ^ (col & 1))            # 0 1 | 1 0 | 0 1 ..
            if p & 1:
                r ^= 1                      # the pair before ended on its row 0 (col 7: k=1 -> r=0): start the next on ITS row 0? see assert
            m = 2 * p + r
            if kind == "zigzag2n":
                m, n = n, m
        out.append((m, n))
    assert len(set(out)) == 64
    if kind.startswith("zigzag2"):
        assert all((a[0] != b[0]) + (a[1] != b[1]) == 1 for a, b in zip(out, out[1:])), out
    return out


def body16(kind, bounded=True):
    o = []
    for h in (0, 1):
        abase = AN if (h and bounded) else A0
        for m, n in order16(kind):
            c = 4 * (8 * m + n)
            o.append(f"v_mfma_f32_16x16x32_bf16 a[{c}:{c + 3}], {frag(B0, n)}, {frag(abase, m)}, a[{c}:{c + 3}]")
    return o


A1, B1, A1N = 136, 168, 200


def body16_two_sets(kind):
    """As the GEMM streams: TWO fragment sets (k-step 0 and 1 of a K tile), every accumulator tile gets one MFMA from each.
    'ksteps': all 64 tiles on set 0, then all 64 on set 1, zigzag (the streams). 'dependent': tile by tile, k-step 0 then k-step 1
    back to back on the same accumulator (does the hardware forward it instead of reading the register file? at the price of
    both sources changing every step). 'pairs': T0 k0, T1 k0, T0 k1, T1 k1 over zigzag neighbours (accumulator reuse distance 2,
    every second step shares a source)."""
    o = []
    z = order16("zigzag2")
    for h in (0, 1):
        a = ((A0, A1), (AN, A1N))[h]
        seq = []
        if kind == "ksteps":
            seq = [(m, n, 0) for m, n in z] + [(m, n, 1) for m, n in z]
        elif kind == "dependent":
            for m, n in z:
                seq += [(m, n, 0), (m, n, 1)]
        elif kind == "double_snake":
            # T0 k0, T0 k1, T1 k1, T1 k0, T2 k0, T2 k1, ..: every step shares the accumulator (k changes) or one source (the tile
            # changes along the zigzag, k stays)
            for i, (m, n) in enumerate(z):
                seq += [(m, n, 0), (m, n, 1)] if i % 2 == 0 else [(m, n, 1), (m, n, 0)]
        elif kind == "pairs":
            for i in range(0, 64, 2):
                (m0, n0), (m1, n1) = z[i], z[i + 1]
                seq += [(m0, n0, 0), (m1, n1, 0), (m0, n0, 1), (m1, n1, 1)]
        for m, n, ks in seq:
            c = 4 * (8 * m + n)
            o.append(f"v_mfma_f32_16x16x32_bf16 a[{c}:{c + 3}], {frag((B0, B1)[ks], n)}, {frag(a[ks], m)}, a[{c}:{c + 3}]")
    return o


def body32(kind):
    """16 accumulator tiles of 32x32 (16 registers each); fragments 0..3 are k-step 0, 4..7 k-step 1."""
    o = []
    for h in (0, 1):
        abase = AN if h else A0
        for ks in (0, 1):
            for i in range(16):
                m, n = i >> 2, i & 3
                if kind == "m_snake" and m & 1:
                    n = 3 - n
                c = 16 * (4 * m + n)
                o.append(f"v_mfma_f32_32x32x16_bf16 a[{c}:{c + 15}], {frag(B0, 4 * ks + n)}, {frag(abase, 4 * ks + m)}, a[{c}:{c + 15}]")
    return o


def body_mx(kind, fmt):
    """f8f6f4 16x16x128: fp8 fragments are 8 registers (A v40.., B v104.., -A v168..), fp4 ones 4 (the bf16 map, cbsz:4 blgp:4)."""
    o = []
    w = 8 if fmt == "fp8" else 4
    a0, b0, an = (40, 104, 168) if fmt == "fp8" else (A0, B0, AN)
    mod = "" if fmt == "fp8" else " cbsz:4 blgp:4"
    for h in (0, 1):
        abase = an if h else a0
        for m, n in order16(kind):
            c = 4 * (8 * m + n)
            o.append(f"v_mfma_f32_16x16x128_f8f6f4 a[{c}:{c + 3}], v[{b0 + w * n}:{b0 + w * n + w - 1}], v[{abase + w * m}:{abase + w * m + w - 1}], a[{c}:{c + 3}]{mod}")
    return o


VARIANTS = [("16x16x32 m-major (the streams' order)", body16("m")), ("16x16x32 m-major snake", body16("m_snake")),
            ("16x16x32 n-major", body16("n")), ("16x16x32 n-major snake", body16("n_snake")),
            ("16x16x32 2x2 blocks, one source changes per step", body16("block2")),
            ("16x16x32 zigzag over row pairs, one source changes per step", body16("zigzag2")),
            ("16x16x32 zigzag over column pairs", body16("zigzag2n")),
            ("16x16x32 diagonal (no shared source)", body16("diag")),
            ("32x32x16 m-major", body32("m")), ("32x32x16 m-major snake", body32("m_snake")),
            ("16x16x32 m-major, growing accumulators", body16("m", bounded=False)),
            ("2sets: 16x16x32 zigzag, k-step 0 then k-step 1 (the streams)", body16_two_sets("ksteps")),
            ("2sets: 16x16x32 tile by tile, both k-steps back to back on one accumulator", body16_two_sets("dependent")),
            ("2sets: 16x16x32 double snake: T0k0 T0k1 T1k1 T1k0 .. (each step shares the accumulator or one source)", body16_two_sets("double_snake")),
            ("2sets: 16x16x32 zigzag neighbours interleaved, accumulator reuse distance 2", body16_two_sets("pairs")),
            ("fp8 16x16x128 n-major (the fp8 stream's order in rounds 2 - 3)", body_mx("n", "fp8")), ("fp8 16x16x128 zigzag over column pairs", body_mx("zigzag2n", "fp8")),
            ("fp8 16x16x128 diagonal (no shared source)", body_mx("diag", "fp8")),
            ("fp4 16x16x128 m-major", body_mx("m", "fp4")), ("fp4 16x16x128 zigzag over row pairs", body_mx("zigzag2", "fp4")),
            ("fp4 16x16x128 diagonal (no shared source)", body_mx("diag", "fp4"))]


def kernel(idx, lines, fmt="bf16"):
    pre = ["s_mov_b32 s64, %[p0]", "s_mov_b32 s65, %[p1]", "s_mov_b32 s66, %[iters]"]
    nload = 32 if fmt in ("fp8", "2sets") else 16
    for i in range(nload):
        base = A0 + 4 * i          # A fragments then B fragments: v40..v103 (fp8: v40..v167)
        pre += [f"global_load_dwordx4 v[{base}:{base + 3}], %[off], s[64:65]", "s_add_u32 s64, s64, 0x1000", "s_addc_u32 s65, s65, 0"]
    pre += [f"v_accvgpr_write_b32 a{i}, 0" for i in range(256)]
    pre += ["s_waitcnt vmcnt(0)"]
    if fmt == "fp8":
        # random bytes hold NaNs (S.1111.111): clear the exponent's top bit; -A flips every byte's sign
        pre += [f"v_and_b32 v{40 + i}, 0xbfbfbfbf, v{40 + i}" for i in range(128)]
        pre += [f"v_xor_b32 v{168 + i}, 0x80808080, v{40 + i}" for i in range(64)]
    elif fmt == "fp4":
        pre += [f"v_xor_b32 v{AN + i}, 0x88888888, v{A0 + i}" for i in range(32)]
    elif fmt == "2sets":
        # loaded: v40..v167 = A0, B0, (scratch), A1; B1 = a copy-free reuse of the scratch block's neighbours is not possible, so
        # move: the third block loaded (v104..135) becomes B1 (v168..199), then the negated sets are derived
        pre += [f"v_mov_b32 v{B1 + i}, v{104 + i}" for i in range(32)]
        pre += [f"v_xor_b32 v{AN + i}, 0x80008000, v{A0 + i}" for i in range(32)]
        pre += [f"v_xor_b32 v{A1N + i}, 0x80008000, v{A1 + i}" for i in range(32)]
    else:
        pre += [f"v_xor_b32 v{AN + i}, 0x80008000, v{A0 + i}" for i in range(32)]
    pre += ["s_nop 4", "s_memtime s[68:69]", "s_memrealtime s[70:71]", "s_waitcnt lgkmcnt(0)", "1:"]
    post = ["s_sub_u32 s66, s66, 1", "s_cmp_lg_u32 s66, 0", "s_cbranch_scc1 1b", "s_memtime s[72:73]", "s_memrealtime s[74:75]", "s_waitcnt lgkmcnt(0)",
            "s_sub_u32 %[dc], s72, s68", "s_sub_u32 %[dr], s74, s70", "s_nop 15", "s_nop 15", "v_accvgpr_read_b32 %[r], a0"]
    text = "".join(f'      "{l}\\n\\t"\n' for l in pre + lines + post)
    clob = ", ".join(f'"{c}"' for c in ["memory", "scc"] + [f"s{i}" for i in range(64, 76)] + [f"v{i}" for i in range(40, 232)] + [f"a{i}" for i in range(256)])
    return f"""__global__ __launch_bounds__(256) void kern{idx}(const char* __restrict__ src, float* __restrict__ sink, int iters) {{
  const unsigned off = threadIdx.x * 16u;
  const unsigned long long p = (unsigned long long)src;
  const unsigned p0 = __builtin_amdgcn_readfirstlane((unsigned)p), p1 = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  float r;
  unsigned dc, dr;
  asm volatile(
{text}      : [r] "=v"(r), [dc] "=s"(dc), [dr] "=s"(dr)
      : [p0] "s"(p0), [p1] "s"(p1), [iters] "s"(iters), [off] "v"(off)
      : {clob});
  if (r == 12345.678f) sink[0] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) sink[2] = (float)dc / (float)dr * 100.0f;     // shader clock in MHz (s_memrealtime: 100 MHz)
}}
"""


MAIN = r"""
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  const int rounds = argc > 2 ? atoi(argv[2]) : 3;
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  std::vector<unsigned short> h(8192 * 8);      // 128 KiB: up to 32 fragment loads of 4 KiB per workgroup
  srand(7);
  for (auto& x : h) {     // bf16 uniform [-1, 1)
    float f = (float)rand() / 2147483648.f * 2.f - 1.f;
    unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16);
  }
  char* src; float* sink;
  CK(hipMalloc(&src, h.size() * 2)); CK(hipMalloc(&sink, 16));
  CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double mac_per_pass_wave_bf16 = 128.0 * 16 * 16 * 32;
  const int it = 10000;
  printf("{\"cus\": %d, \"rows\": [\n", cus);
  for (int r = 0; r < rounds; ++r)
    for (int which = 0; which < NW; ++which) {
      double t_total = 0; long launches = 0;
      CK(hipEventRecord(e0));
      while (true) {
        for (int q = 0; q < 5; ++q) { launch(which, cus, src, sink, it); ++launches; }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        t_total = ms * 1e-3;
        if (t_total >= secs) break;
      }
      const double mac_per_pass_wave = mac_per_pass_wave_bf16 * ((names[which][0] == 'f') ? 4.0 : (names[which][0] == '2') ? 2.0 : 1.0);     // 16x16x128: four times the MACs; two sets: 256 MFMAs per pass
      const double peak_per_clk = (names[which][0] == 'f') ? (names[which][2] == '4' ? 4096.0 : 2048.0) : 1024.0;
      const double flops = 2.0 * mac_per_pass_wave * it * 4.0 * cus * launches / t_total;
      float mhz = 0; CK(hipMemcpy(&mhz, sink + 2, 4, hipMemcpyDeviceToHost));
      // busy: the rate over what 4 SIMDs x 1024 FLOP per clock would give at the clock the last launch ran at
      printf("  {\"round\": %d, \"variant\": \"%s\", \"seconds\": %.3f, \"tflops\": %.1f, \"clock_mhz\": %.0f, \"mfma_busy\": %.3f}%s\n", r, names[which], t_total, flops / 1e12,
             mhz, flops / (4.0 * cus * peak_per_clk * mhz * 1e6),
             (r == rounds - 1 && which == NW - 1) ? "" : ",");
      fflush(stdout);
    }
  printf("]}\n");
  return 0;
}
"""


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "native", "mfma_order_power.hip")
    with open(out, "w") as f:
        f.write("// GENERATED by scripts/exp/gen_mfma_order_exp.py -- do not edit.\n#include <hip/hip_runtime.h>\n#include <cstdio>\n"
                "#include <cstdlib>\n#include <cstring>\n#include <vector>\n\n")
        for i, (name, lines) in enumerate(VARIANTS):
            f.write(kernel(i, lines, "fp8" if name.startswith("fp8") else "fp4" if name.startswith("fp4") else "2sets" if name.startswith("2sets") else "bf16"))
        f.write(f"#define NW {len(VARIANTS)}\nstatic const char* names[NW] = {{" + ", ".join('"%s"' % n for n, _ in VARIANTS) + "};\n")
        f.write("static void launch(int w, int cus, const char* src, float* sink, int it) {\n  switch (w) {\n")
        for i in range(len(VARIANTS)):
            f.write(f"    case {i}: hipLaunchKernelGGL(kern{i}, dim3(cus), dim3(256), 0, 0, src, sink, it); break;\n")
        f.write("  }\n}\n")
        f.write(MAIN)
    print(out)


if __name__ == "__main__":
    main()
