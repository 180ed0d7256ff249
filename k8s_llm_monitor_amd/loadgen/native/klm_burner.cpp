// klm_burner: standalone MI355X load generator (no Python, no PyTorch): the MFMA GEMM
// kernels and HBM kernels of ops/csrc behind a duty-cycle loop, for pods that carry only
// the ROCm runtime. Same control law as loadgen/burner.py.
//
// No reference counterpart (BASELINE.json north_star item 4).
//
//   klm_burner --kind gemm_bf16|gemm_fp8|gemm_fp4|hbm_copy|hbm_read|hbm_fill|spin
//              [--size 8192] [--duty 100] [--period-ms 100] [--seconds 10]
//              [--hbm-gib G]   also allocate and touch G GiB (HBM exhaustion injector)
//              [--device 0]
//   klm_burner --selftest      no GPU: the window planner and the JSON writers on fixed inputs (sanitizer runs)
// Output: one JSON line per second {"t","kernels","duty","tflops"|"TBps"} and a final summary.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "klm_ops.h"

#define HIPCHECK(x)                                                                   \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("{\"error\":\"%s: %s\"}\n", #x, hipGetErrorString(e_));                  \
      fflush(stdout);                                                                 \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)

static volatile sig_atomic_t g_stop = 0;
static void on_signal(int) { g_stop = 1; }
static double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// xorshift fill on the host would take seconds for 8192^2; fill on the device instead
__global__ void fill_bf16_uniform(uint16_t* p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const float f = (float)(x & 0xFFFF) / 32768.0f - 1.0f;   // uniform [-1, 1): zero operands read high (DVFS)
    uint32_t u;
    memcpy(&u, &f, 4);
    p[i] = (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
  }
}
// e4m3 bit patterns with exponent <= 6 (|v| < 1) and any mantissa, random sign: finite, no NaN
__global__ void fill_fp8_uniform(uint8_t* p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2246822519u ^ seed;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const uint32_t e = (x >> 8) % 7u, m = x & 7u, s = (x >> 16) & 1u;
    p[i] = (uint8_t)((s << 7) | (e << 3) | m);
  }
}

__global__ void fill_bytes_random(uint8_t* p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2246822519u ^ seed;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    p[i] = (uint8_t)(x >> 11);
  }
}

struct Args {
  std::string kind = "gemm_bf16";
  int size = 8192, device = 0;
  double duty = 100, period_ms = 100, seconds = 10, hbm_gib = 0;
};

static const char* const KINDS[] = {"gemm_bf16", "gemm_fp8", "gemm_fp4", "hbm_copy", "hbm_read", "hbm_fill", "spin"};
static bool known_kind(const std::string& k) {
  for (const char* x : KINDS)
    if (k == x) return true;
  return false;
}
// a string inside a JSON value: quotes, backslashes and control bytes escaped
static std::string json_escape(const std::string& in) {
  std::string out;
  for (unsigned char c : in) {
    if (c == '"' || c == '\\') { out += '\\'; out += (char)c; }
    else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); out += b; }
    else out += (char)c;
  }
  return out;
}
// a dense row pitch (bytes) the memory channels camp on: a multiple of 2 KiB that is no power of two (the same rule as
// ops.camping_pitch; measured in scripts/exp/gemm_stride_sweep.py, gemm_pitch_check.py)
static bool camping_pitch(long nbytes) { return nbytes >= 4096 && nbytes % 2048 == 0 && (nbytes & (nbytes - 1)) != 0; }
// kernels per busy window: the whole number whose run time is closest to duty * period, at least 1 when
// duty > 0 (the same law as loadgen/burner.py::plan_window)
static int plan_window(double duty, double period_s, double kernel_s) {
  if (duty <= 0 || kernel_s <= 0) return 0;
  return std::max(1, (int)std::lround(std::min(1.0, duty / 100.0) * period_s / kernel_s));
}
static void print_report(double wall, long kernels, double busy_s, double flops, double bytes) {
  printf("{\"t\":%.3f,\"kernels\":%ld,\"duty\":%.2f", wall, kernels, std::min(100.0, busy_s / wall * 100.0));
  if (flops) printf(",\"tflops\":%.2f", flops * kernels / std::max(busy_s, 1e-9) / 1e12);
  if (bytes) printf(",\"TBps\":%.3f", bytes * kernels / std::max(busy_s, 1e-9) / 1e12);
  printf("}\n");
  fflush(stdout);
}
static void print_summary(const Args& a, double kernel_s, long kernels, double wall, double busy_s, double flops, double bytes, double held) {
  printf("{\"kind\":\"%s\",\"size\":%d,\"duty_target\":%.2f,\"period_ms\":%.2f,\"kernel_ms\":%.4f,\"kernels\":%ld,"
         "\"wall_s\":%.3f,\"duty_achieved\":%.3f",
         json_escape(a.kind).c_str(), a.size, a.duty, a.period_ms, kernel_s * 1e3, kernels, wall, std::min(100.0, busy_s / wall * 100.0));
  if (flops) printf(",\"tflops_busy\":%.2f,\"tflops_avg\":%.2f", flops * kernels / std::max(busy_s, 1e-9) / 1e12, flops * kernels / wall / 1e12);
  if (bytes) printf(",\"TBps_busy\":%.3f,\"TBps_avg\":%.3f", bytes * kernels / std::max(busy_s, 1e-9) / 1e12, bytes * kernels / wall / 1e12);
  if (held > 0) printf(",\"hbm_hold_gib\":%.3f", held / (double)((size_t)1 << 30));
  printf("}\n");
  fflush(stdout);
}

// --selftest: no device. Prints the planner's answers for a fixed table (tests compare them with the Python
// planner) and one report and one summary line built from fixed numbers, with a hostile kind string.
static int selftest() {
  const double duties[] = {0, 1, 25, 50, 75, 99.5, 100}, kernels_s[] = {20e-6, 730e-6, 2.1e-3, 0.25};
  printf("{\"selftest\":\"plan\",\"period_s\":0.1,\"rows\":[");
  bool first = true;
  for (double d : duties)
    for (double k : kernels_s) {
      printf("%s[%.2f,%.6g,%d]", first ? "" : ",", d, k, plan_window(d, 0.1, k));
      first = false;
    }
  printf("]}\n");
  // the library's default dispatch for a square problem of each size on 256 CUs (no device needed: the CU count is given):
  // [size, bf16 kind, fp8 kind, fp4 kind], 0 none / 1 persistent stream / 2 ring kernel 128 x 128 / 3 ring kernel 256 x 128
  printf("{\"selftest\":\"gemm_pick\",\"cus\":256,\"rows\":[");
  const int sizes[] = {256, 1024, 2048, 2560, 3072, 4096, 6144, 8192, 12288};
  for (size_t i = 0; i < sizeof sizes / sizeof *sizes; ++i)
    printf("%s[%d,%d,%d,%d]", i ? "," : "", sizes[i], klm_gemm_pick(sizes[i], sizes[i], 2 * sizes[i], 256), klm_gemm_pick(sizes[i], sizes[i], sizes[i], 256),
           klm_gemm_pick(sizes[i], sizes[i], sizes[i] / 2, 256));
  printf("]}\n");
  printf("{\"selftest\":\"camping_pitch\",\"rows\":[");
  const long pitches[] = {512, 4096, 6144, 8192, 12288, 16384, 20480, 24064, 24576, 28672, 32768, 49152};
  for (size_t i = 0; i < sizeof pitches / sizeof *pitches; ++i) printf("%s[%ld,%d]", i ? "," : "", pitches[i], camping_pitch(pitches[i]) ? 1 : 0);
  printf("]}\n");
  print_report(2.0, 2700, 1.97, 2.0 * 8192.0 * 8192.0 * 8192.0, 0.0);
  print_report(2.0, 1400, 2.0, 0.0, 4294967296.0);
  Args a;
  a.kind = "gemm\"bf16\\\n";
  print_summary(a, 730e-6, 2700, 2.0, 1.97, 2.0 * 8192.0 * 8192.0 * 8192.0, 0.0, 3.0 * (double)((size_t)1 << 30));
  return 0;
}

int main(int argc, char** argv) {
  Args a;
  for (int i = 1; i < argc; ++i) {
    std::string k = argv[i];
    auto val = [&]() -> const char* {
      if (i + 1 >= argc) { printf("{\"error\":\"missing value for %s\"}\n", k.c_str()); exit(1); }
      return argv[++i];
    };
    if (k == "--kind") a.kind = val();
    else if (k == "--size") a.size = atoi(val());
    else if (k == "--duty") a.duty = atof(val());
    else if (k == "--period-ms") a.period_ms = atof(val());
    else if (k == "--seconds") a.seconds = atof(val());
    else if (k == "--hbm-gib") a.hbm_gib = atof(val());
    else if (k == "--device") a.device = atoi(val());
    else if (k == "--selftest") return selftest();
    else {
      fprintf(stderr, "usage: klm_burner --kind gemm_bf16|gemm_fp8|gemm_fp4|hbm_copy|hbm_read|hbm_fill|spin [--size 8192] [--duty 100] "
                      "[--period-ms 100] [--seconds 10] [--hbm-gib G] [--device 0] | --selftest\n");
      return (k == "--help" || k == "-h") ? 0 : 1;
    }
  }
  if (!known_kind(a.kind)) {      // before any HIP call: a typo must not need a GPU to be reported
    printf("{\"error\":\"unknown kind %s\"}\n", json_escape(a.kind).c_str());
    return 1;
  }
  if (a.device < 0 || a.hbm_gib < 0 || a.hbm_gib > 1024) {
    printf("{\"error\":\"bad arguments (device >= 0, hbm-gib 0..1024)\"}\n");
    return 1;
  }
  if (a.duty < 0 || a.duty > 100 || a.period_ms <= 0 || a.seconds <= 0 || a.size <= 0 || a.size % 256) {
    printf("{\"error\":\"bad arguments (size must be a multiple of 256, duty 0..100)\"}\n");
    return 1;
  }
  signal(SIGINT, on_signal);
  signal(SIGTERM, on_signal);
  HIPCHECK(hipSetDevice(a.device));
  hipStream_t s;
  HIPCHECK(hipStreamCreate(&s));

  const bool gemm = a.kind == "gemm_bf16" || a.kind == "gemm_fp8" || a.kind == "gemm_fp4";
  const bool fp8 = a.kind == "gemm_fp8", fp4 = a.kind == "gemm_fp4";
  const bool hbm = a.kind.rfind("hbm_", 0) == 0;
  void *A = nullptr, *B = nullptr, *C = nullptr, *X = nullptr, *Y = nullptr, *sink = nullptr;
  double flops = 0, bytes = 0;
  const size_t n2 = (size_t)a.size * a.size;
  // Row pitch of A and B: dense, unless the hand-scheduled stream takes the shape and the dense pitch is one the memory channels
  // camp on (a multiple of 2 KiB that is no power of two: 8192 x 8192 x 12288 bf16 runs at 1380 TFLOP/s dense, 1617 with 256
  // bytes of padding per row; docs/KERNELS.md "row pitch"). The operands are the burner's own, so it may lay them out.
  const int row_bytes = fp4 ? a.size / 2 : fp8 ? a.size : a.size * 2;
  const bool stream_ok = gemm && row_bytes % 256 == 0 && row_bytes >= 512;
  const int pitch = row_bytes + ((stream_ok && camping_pitch(row_bytes)) ? 256 : 0);
  if (gemm) {
    const size_t abytes = (size_t)a.size * pitch;
    HIPCHECK(hipMalloc(&A, abytes));
    HIPCHECK(hipMalloc(&B, abytes));
    HIPCHECK(hipMalloc(&C, n2 * 2));
    if (fp4) {   // every byte is two valid e2m1 codes: random bytes are random fp4 operands
      hipLaunchKernelGGL(fill_bytes_random, dim3(2048), dim3(256), 0, s, (uint8_t*)A, abytes, 1u);
      hipLaunchKernelGGL(fill_bytes_random, dim3(2048), dim3(256), 0, s, (uint8_t*)B, abytes, 2u);
    } else if (fp8) {
      hipLaunchKernelGGL(fill_fp8_uniform, dim3(2048), dim3(256), 0, s, (uint8_t*)A, abytes, 1u);
      hipLaunchKernelGGL(fill_fp8_uniform, dim3(2048), dim3(256), 0, s, (uint8_t*)B, abytes, 2u);
    } else {
      hipLaunchKernelGGL(fill_bf16_uniform, dim3(2048), dim3(256), 0, s, (uint16_t*)A, abytes / 2, 1u);
      hipLaunchKernelGGL(fill_bf16_uniform, dim3(2048), dim3(256), 0, s, (uint16_t*)B, abytes / 2, 2u);
    }
    flops = 2.0 * a.size * (double)a.size * a.size;
  } else if (hbm) {
    const size_t n = (size_t)2 << 30;
    HIPCHECK(hipMalloc(&X, n));
    HIPCHECK(hipMalloc(&sink, 16));
    HIPCHECK(hipMemsetAsync(sink, 0, 16, s));
    klm_hbm_fill(X, n, 0x5A5A5A5Au, s);
    if (a.kind == "hbm_copy") {
      HIPCHECK(hipMalloc(&Y, n));
      bytes = 2.0 * n;
    } else {
      bytes = (double)n;
    }
  }
  // HBM exhaustion injector: hold and touch G GiB in 4 GiB chunks
  std::vector<void*> hold;
  double held = 0;
  if (a.hbm_gib > 0) {
    size_t left = (size_t)(a.hbm_gib * (double)((size_t)1 << 30)) & ~(size_t)0xF;
    while (left) {
      const size_t n = std::min(left, (size_t)4 << 30);
      void* p = nullptr;
      if (hipMalloc(&p, n) != hipSuccess) { (void)hipGetLastError(); break; }
      klm_hbm_fill(p, n, 0xA5A5A5A5u, s);
      hold.push_back(p);
      held += (double)n;
      left -= n;
    }
    HIPCHECK(hipStreamSynchronize(s));
    printf("{\"hbm_hold_gib\":%.3f}\n", held / (double)((size_t)1 << 30));
    fflush(stdout);
  }

  // scratch for the streams' split of a last partial round: one 256 KiB slot per CU
  void* ws = nullptr;
  size_t ws_bytes = 0;
  int picked = -1;
  if (gemm) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, a.device) == hipSuccess && cus > 0) {
      ws_bytes = (size_t)cus << 18;
      if (hipMalloc(&ws, ws_bytes) != hipSuccess) { (void)hipGetLastError(); ws = nullptr; ws_bytes = 0; }
    }
  }
  auto launch = [&]() -> int {
    // padded operands (a camping pitch): the persistent hand-scheduled streams, which take a row pitch
    if (pitch != row_bytes)
      return fp4 || fp8 ? klm_gemm_mx_pitched(fp8 ? 1 : 2, A, B, C, a.size, a.size, row_bytes, pitch, 1, s)
                        : klm_gemm_bf16_pitched(A, B, C, a.size, a.size, a.size, pitch, 1, s);
    // dense operands: the library's default dispatch (cost model: the persistent stream with its plan, or the eight-wave ring
    // kernel on 128 x 128 / 256 x 128 tiles for sizes of too few 256 x 256 tiles to fill the chip)
    if (gemm) return klm_gemm_auto(fp4 ? 2 : fp8 ? 1 : 0, A, B, C, a.size, a.size, (int)row_bytes, ws, ws_bytes, s, &picked);
    if (a.kind == "hbm_copy") return klm_hbm_copy(Y, X, (size_t)2 << 30, s);
    if (a.kind == "hbm_read") return klm_hbm_read(X, (size_t)2 << 30, sink, s);
    if (a.kind == "hbm_fill") return klm_hbm_fill(X, (size_t)2 << 30, 0x5A5A5A5Au, s);
    return klm_spin(2400000ull, 1024, s);
  };

  // calibrate the per-kernel time
  hipEvent_t c0, c1;
  HIPCHECK(hipEventCreate(&c0));
  HIPCHECK(hipEventCreate(&c1));
  for (int i = 0; i < 3; ++i)
    if (launch() != KLM_OK) { printf("{\"error\":\"kernel launch failed\"}\n"); return 3; }
  HIPCHECK(hipEventRecord(c0, s));
  for (int i = 0; i < 10; ++i) launch();
  HIPCHECK(hipEventRecord(c1, s));
  HIPCHECK(hipEventSynchronize(c1));
  float cal_ms = 0;
  HIPCHECK(hipEventElapsedTime(&cal_ms, c0, c1));
  const double kernel_s = cal_ms * 1e-3 / 10;
  const double period_s = a.period_ms * 1e-3;
  const bool full = a.duty >= 100.0;
  // closed loop: every retired window updates the per-kernel estimate the next one is sized by
  double k_est = kernel_s;
  auto plan = [&]() { return plan_window(a.duty, period_s, k_est); };

  struct Win { hipEvent_t e0, e1; int n; };
  std::deque<Win> pending;
  std::vector<Win> pool;
  auto get_win = [&]() {
    if (!pool.empty()) { Win w = pool.back(); pool.pop_back(); return w; }
    Win w{};
    HIPCHECK(hipEventCreate(&w.e0));
    HIPCHECK(hipEventCreate(&w.e1));
    return w;
  };
  double busy_s = 0;
  long kernels = 0;
  auto retire = [&](bool block) {
    if (block && !pending.empty()) HIPCHECK(hipEventSynchronize(pending.front().e1));
    while (!pending.empty() && hipEventQuery(pending.front().e1) == hipSuccess) {
      float ms = 0;
      HIPCHECK(hipEventElapsedTime(&ms, pending.front().e0, pending.front().e1));
      busy_s += ms * 1e-3;
      kernels += pending.front().n;
      if (pending.front().n) k_est = 0.7 * k_est + 0.3 * ms * 1e-3 / pending.front().n;
      pool.push_back(pending.front());
      pending.pop_front();
    }
    (void)hipGetLastError();  // hipEventQuery's "not ready" is not an error
  };

  const double t_start = now_s(), t_end = t_start + a.seconds;
  double next_report = t_start + 1.0;
  long k = 0;
  while (!g_stop && now_s() < t_end) {
    if (!full) {
      const double p0 = t_start + k * period_s, now = now_s();
      if (p0 > now) std::this_thread::sleep_for(std::chrono::duration<double>(std::min(p0 - now, t_end - now)));
      if (now_s() >= t_end) break;
    }
    const int n_per = plan();
    if (n_per) {
      Win w = get_win();
      w.n = n_per;
      HIPCHECK(hipEventRecord(w.e0, s));
      for (int i = 0; i < n_per; ++i) launch();
      HIPCHECK(hipEventRecord(w.e1, s));
      pending.push_back(w);
      if (full) retire(pending.size() >= 4);
      else { HIPCHECK(hipEventSynchronize(w.e1)); retire(false); }
    } else {
      std::this_thread::sleep_for(std::chrono::duration<double>(period_s));
    }
    ++k;
    const double now = now_s();
    if (now >= next_report) {
      next_report += 1.0;
      print_report(now - t_start, kernels, busy_s, flops, bytes);
    }
  }
  while (!pending.empty()) retire(true);
  print_summary(a, kernel_s, kernels, now_s() - t_start, busy_s, flops, bytes, held);
  for (void* p : hold) (void)hipFree(p);
  for (void* p : {A, B, C, X, Y, sink, ws}) if (p) (void)hipFree(p);
  return 0;
}
