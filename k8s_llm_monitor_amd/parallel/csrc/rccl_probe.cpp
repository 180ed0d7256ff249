// klm_rccl_probe: RCCL all-reduce bandwidth probe over xGMI for MI355X nodes.
//
// No reference counterpart (the reference has no collective call site);
// BASELINE.json north_star item 4 / SURVEY.md 2.4 N5: "ncclAllReduce sum at 1, 2, 4, 8 GPUs
// over xGMI, algBW and busBW per size, JSON the server exposes and the dashboard plots".
//
// One process, one communicator per device (ncclCommInitAll), all ranks driven from this
// thread inside ncclGroupStart/End, IN PLACE (send == recv, as a gradient all-reduce runs;
// --out-of-place for the other form). Per message size: `warmup` untimed rounds, then `iters`
// rounds timed with a hipEvent pair on EVERY rank's stream between two cross-device syncs; the
// reported time is the slowest rank's (`time_us` = `time_us_max_rank`; rank 0's own beside it).
// Output: JSON lines on stdout (parallel/probe.py parses them):
//   {"type":"header",...}  {"type":"point",...}*  {"type":"summary",...}   or {"type":"error",...}
//
// busBW = algBW * 2(n-1)/n (nccl-tests convention). n = 1 has no link traffic, and an IN-PLACE all-reduce on
// a one-rank communicator is a no-op (bytes / time of nothing): at n = 1 the probe therefore always runs OUT OF
// PLACE, which RCCL serves with a real device-local copy; the curve is labelled "local_copy" and carries algBW
// with busBW 0 ("no traffic; report local copy rate separately").
//
// Every point is held against a physical bound before it is printed: nothing on this card moves more than
// 8 TB/s locally, and no all-reduce can beat the full mesh ((n-1) links x 153 GB/s, +10 % for the link rate's
// own uncertainty). A point above its bound gets check "implausible" (the parser marks the curve unverified).
//
// Data check: each rank's buffer holds (rank + 1) before the first all-reduce of a size; the
// sum must be n(n+1)/2 in EVERY element of every rank's buffer (exact in bf16 and fp32 for
// n <= 8): a compare kernel counts the elements that differ. The timed rounds then reduce the
// buffer onto itself again and again; the values grow (x n per round) and stay finite for the
// default round counts, which is all a bandwidth figure needs.
#include <hip/hip_runtime.h>
#include <rccl.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// the hash of the sources this binary was built from (build.py passes KLM_SRC_HASH and finds the marker to decide staleness)
#ifndef KLM_SRC_HASH
#define KLM_SRC_HASH "unhashed"
#endif
extern "C" __attribute__((used)) const char klm_src_hash_marker[] = "KLM_SRC_HASH=" KLM_SRC_HASH;

#define HIPCHECK(x)                                                                        \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      printf("{\"type\":\"error\",\"error\":\"%s: %s\"}\n", #x, hipGetErrorString(e_));    \
      fflush(stdout);                                                                      \
      exit(2);                                                                             \
    }                                                                                      \
  } while (0)
#define NCCLCHECK(x)                                                                       \
  do {                                                                                     \
    ncclResult_t r_ = (x);                                                                 \
    if (r_ != ncclSuccess) {                                                               \
      printf("{\"type\":\"error\",\"error\":\"%s: %s\"}\n", #x, ncclGetErrorString(r_));   \
      fflush(stdout);                                                                      \
      exit(3);                                                                             \
    }                                                                                      \
  } while (0)

struct Args {
  int gpus = 1;
  std::string dtype = "bf16";
  size_t min_bytes = 1024, max_bytes = (size_t)1 << 30;
  int factor = 4, warmup = 5, iters = 20;
  bool check = true, in_place = true, selftest = false, selftest_bounds = false;
};

static bool parse(int argc, char** argv, Args& a) {
  for (int i = 1; i < argc; ++i) {
    std::string k = argv[i];
    auto next = [&](const char* name) -> const char* {
      if (i + 1 >= argc) {
        printf("{\"type\":\"error\",\"error\":\"missing value for %s\"}\n", name);
        exit(1);
      }
      return argv[++i];
    };
    if (k == "--gpus") a.gpus = atoi(next("--gpus"));
    else if (k == "--dtype") a.dtype = next("--dtype");
    else if (k == "--min") a.min_bytes = strtoull(next("--min"), nullptr, 10);
    else if (k == "--max") a.max_bytes = strtoull(next("--max"), nullptr, 10);
    else if (k == "--factor") a.factor = atoi(next("--factor"));
    else if (k == "--warmup") a.warmup = atoi(next("--warmup"));
    else if (k == "--iters") a.iters = atoi(next("--iters"));
    else if (k == "--no-check") a.check = false;
    else if (k == "--out-of-place") a.in_place = false;
    else if (k == "--selftest") a.selftest = true;
    else if (k == "--selftest-bounds") a.selftest = a.selftest_bounds = true;
    else if (k == "--help" || k == "-h") return false;
    else {
      printf("{\"type\":\"error\",\"error\":\"unknown argument %s\"}\n", k.c_str());
      exit(1);
    }
  }
  return true;
}

// value -> element bit pattern
static uint16_t bf16_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)(u >> 16);  // small integers are exact in bf16
}
static float bf16_to_float(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

__global__ void fill16(uint16_t* p, size_t n, uint16_t v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill32(float* p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// elements of p[0..n) whose bit pattern is not `want` -> *bad (one atomic per thread that found any)
template <typename T>
__global__ void count_mismatch(const T* p, size_t n, T want, unsigned long long* bad) {
  unsigned long long mine = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) mine += p[i] != want;
  if (mine) atomicAdd(bad, mine);
}

static std::string json_escape(const std::string& in) {
  std::string out;
  for (unsigned char c : in) {
    if (c == '"' || c == '\\') { out += '\\'; out += (char)c; }
    else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); out += b; }
    else out += (char)c;
  }
  return out;
}
// message sizes of the sweep: min, min*f, ... below max, then max itself; each a whole number of elements
static std::vector<size_t> sweep_sizes(size_t min_bytes, size_t max_bytes, int factor, size_t esz) {
  std::vector<size_t> sizes;
  for (size_t b = min_bytes; b < max_bytes; b *= (size_t)factor) {
    sizes.push_back(b - b % esz);
    if (b > max_bytes / (size_t)factor) break;      // the next product would pass max (or overflow)
  }
  sizes.push_back(max_bytes - max_bytes % esz);
  return sizes;
}
// the physical bounds of a point, GB/s: local copy rate at n = 1, bus bandwidth of the full mesh beyond
static const double kLocalBoundGBps = 8000.0, kLinkGBps = 153.0, kLinkSlack = 1.1;
static double bound_GBps(int n) { return n > 1 ? (n - 1) * kLinkGBps * kLinkSlack : kLocalBoundGBps; }
static bool plausible(int n, double alg, double bus) { return (n > 1 ? bus : alg) <= bound_GBps(n); }

static void print_header(const Args& a, int n, int ver, const char* device, const char* arch) {
  printf("{\"type\":\"header\",\"n_gpus\":%d,\"dtype\":\"%s\",\"op\":\"allreduce\",\"warmup\":%d,\"iters\":%d,"
         "\"rccl_version\":%d,\"device\":\"%s\",\"arch\":\"%s\",\"min_bytes\":%zu,\"max_bytes\":%zu,\"factor\":%d,"
         "\"in_place\":%s,\"mode\":\"%s\",\"bound_GBps\":%.1f,\"timing\":\"event pair per rank, max over ranks\",\"check\":\"%s\"}\n",
         n, json_escape(a.dtype).c_str(), a.warmup, a.iters, ver, json_escape(device).c_str(), json_escape(arch).c_str(),
         a.min_bytes, a.max_bytes, a.factor, a.in_place ? "true" : "false", n > 1 ? "allreduce" : "local_copy", bound_GBps(n),
         a.check ? "every element, device compare" : "off");
  fflush(stdout);
}
struct Point {
  double sec = 0, sec_rank0 = 0, alg = 0, bus = 0;
};
static Point make_point(size_t bytes, int n, int iters, float ms_max, float ms_rank0) {
  Point p;
  p.sec = (double)ms_max * 1e-3 / iters;
  p.sec_rank0 = (double)ms_rank0 * 1e-3 / iters;
  p.alg = p.sec > 0 ? bytes / p.sec / 1e9 : 0.0;
  p.bus = p.alg * (n > 1 ? 2.0 * (n - 1) / n : 0.0);
  return p;
}
static void print_point(const Args& a, int n, size_t bytes, const Point& p, const char* check, size_t checked, unsigned long long mismatched) {
  if (!plausible(n, p.alg, p.bus) && strcmp(check, "MISMATCH") != 0) check = "implausible";
  printf("{\"type\":\"point\",\"n_gpus\":%d,\"bytes\":%zu,\"dtype\":\"%s\",\"op\":\"allreduce\",\"time_us\":%.3f,"
         "\"time_us_max_rank\":%.3f,\"time_us_rank0\":%.3f,\"algbw_GBps\":%.4f,\"busbw_GBps\":%.4f,\"check\":\"%s\","
         "\"elements_checked\":%zu,\"elements_mismatched\":%llu}\n",
         n, bytes, json_escape(a.dtype).c_str(), p.sec * 1e6, p.sec * 1e6, p.sec_rank0 * 1e6, p.alg, p.bus, check, checked, mismatched);
  fflush(stdout);
}
static void print_summary(const Args& a, int n, double peak_bus, double peak_alg, size_t peak_bytes) {
  printf("{\"type\":\"summary\",\"n_gpus\":%d,\"dtype\":\"%s\",\"peak_busbw_GBps\":%.4f,\"peak_algbw_GBps\":%.4f,"
         "\"peak_bytes\":%zu,\"ceiling_mesh_GBps\":%.1f,\"ceiling_ring_GBps\":%.1f}\n",
         n, json_escape(a.dtype).c_str(), peak_bus, peak_alg, peak_bytes, n > 1 ? (n - 1) * 153.0 : 0.0, n > 1 ? 153.0 : 0.0);
  fflush(stdout);
}
// --selftest: no device and no RCCL call. The sweep's size list and the three JSON writers on fixed numbers
// (8 ranks, 256 MiB in 1.25 ms on the slowest rank), for the sanitizer run and the parser's tests.
static int selftest(Args a) {
  a.gpus = 8;
  print_header(a, 8, 22705, "AMD \"Instinct\" MI355X\\", "gfx950:sramecc+:xnack-");
  double peak_bus = 0, peak_alg = 0;
  size_t peak_bytes = 0;
  for (size_t bytes : sweep_sizes(a.min_bytes, a.max_bytes, a.factor, a.dtype == "bf16" ? 2 : 4)) {
    const float ms = (float)(20.0 * (0.02 + bytes / 214.7e6));          // 20 us latency + 214.7 GB/s algBW, 20 iters
    const Point p = make_point(bytes, 8, 20, ms, ms * 0.97f);
    if (p.bus > peak_bus) peak_bytes = bytes;
    peak_bus = std::max(peak_bus, p.bus);
    peak_alg = std::max(peak_alg, p.alg);
    print_point(a, 8, bytes, p, "ok", bytes / 2 * 8, 0);
  }
  print_summary(a, 8, peak_bus, peak_alg, peak_bytes);
  if (!a.selftest_bounds) return 0;
  // --selftest-bounds, the bound itself: a one-rank "copy" of 256 MiB in 0.436 us (what an in-place no-op once read as) and an
  // 8-rank point above the mesh ceiling must both come out "implausible"; one just under each bound stays "ok"
  const size_t mib256 = (size_t)256 << 20;
  print_point(a, 1, mib256, make_point(mib256, 1, 1, 0.000436f, 0.000436f), "ok", mib256 / 2, 0);
  print_point(a, 1, mib256, make_point(mib256, 1, 1, (float)(mib256 / 7.9e12 * 1e3), (float)(mib256 / 7.9e12 * 1e3)), "ok", mib256 / 2, 0);
  print_point(a, 8, mib256, make_point(mib256, 8, 1, (float)(mib256 * 1.75 / 1200e9 * 1e3), 0.3f), "ok", mib256 * 4, 0);
  print_point(a, 8, mib256, make_point(mib256, 8, 1, (float)(mib256 * 1.75 / 1150e9 * 1e3), 0.3f), "ok", mib256 * 4, 0);
  return 0;
}

int main(int argc, char** argv) {
  Args a;
  if (!parse(argc, argv, a)) {
    fprintf(stderr, "usage: klm_rccl_probe [--gpus N] [--dtype bf16|fp32] [--min B] [--max B] [--factor F] "
                    "[--warmup W] [--iters I] [--no-check] [--out-of-place] | --selftest | --selftest-bounds\n");
    return 0;
  }
  const bool is_bf16 = a.dtype == "bf16";
  if (!is_bf16 && a.dtype != "fp32") {
    printf("{\"type\":\"error\",\"error\":\"dtype must be bf16 or fp32\"}\n");
    return 1;
  }
  const size_t esz = is_bf16 ? 2 : 4;
  const ncclDataType_t nt = is_bf16 ? ncclBfloat16 : ncclFloat32;
  if (a.gpus < 1 || a.factor < 2 || a.min_bytes < esz || a.max_bytes < a.min_bytes || a.iters < 1 || a.warmup < 0) {
    printf("{\"type\":\"error\",\"error\":\"bad arguments\"}\n");
    return 1;
  }
  if (a.selftest) return selftest(a);
  int have = 0;
  HIPCHECK(hipGetDeviceCount(&have));
  if (have < a.gpus) {
    printf("{\"type\":\"error\",\"error\":\"need %d devices, have %d\"}\n", a.gpus, have);
    return 4;
  }
  const int n = a.gpus;
  if (n == 1) a.in_place = false;      // one rank: in place is a no-op; out of place is the device-local copy the point reports
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) devs[i] = i;
  std::vector<ncclComm_t> comms(n);
  NCCLCHECK(ncclCommInitAll(comms.data(), n, devs.data()));
  int ver = 0;
  ncclGetVersion(&ver);

  std::vector<hipStream_t> streams(n);
  std::vector<void*> send(n), recv(n);
  std::vector<unsigned long long*> bad(n);
  std::vector<hipEvent_t> ev0(n), ev1(n);
  for (int i = 0; i < n; ++i) {
    HIPCHECK(hipSetDevice(i));
    HIPCHECK(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));
    HIPCHECK(hipMalloc(&send[i], a.max_bytes));
    if (a.in_place) recv[i] = send[i];
    else HIPCHECK(hipMalloc(&recv[i], a.max_bytes));
    HIPCHECK(hipMalloc((void**)&bad[i], sizeof(unsigned long long)));
    HIPCHECK(hipEventCreate(&ev0[i]));
    HIPCHECK(hipEventCreate(&ev1[i]));
  }
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, 0));
  print_header(a, n, ver, prop.name, prop.gcnArchName);

  auto sync_all = [&]() {
    for (int i = 0; i < n; ++i) {
      HIPCHECK(hipSetDevice(i));
      HIPCHECK(hipStreamSynchronize(streams[i]));
    }
  };
  auto round = [&](size_t count) {
    NCCLCHECK(ncclGroupStart());
    for (int i = 0; i < n; ++i) NCCLCHECK(ncclAllReduce(send[i], recv[i], count, nt, ncclSum, comms[i], streams[i]));
    NCCLCHECK(ncclGroupEnd());
  };

  const std::vector<size_t> sizes = sweep_sizes(a.min_bytes, a.max_bytes, a.factor, esz);

  double peak_bus = 0, peak_alg = 0;
  size_t peak_bytes = 0;
  for (size_t bytes : sizes) {
    const size_t count = bytes / esz;
    for (int i = 0; i < n; ++i) {
      HIPCHECK(hipSetDevice(i));
      const int grid = (int)std::min<size_t>(2048, (count + 255) / 256);
      if (is_bf16) hipLaunchKernelGGL(fill16, dim3(grid), dim3(256), 0, streams[i], (uint16_t*)send[i], count, bf16_bits((float)(i + 1)));
      else hipLaunchKernelGGL(fill32, dim3(grid), dim3(256), 0, streams[i], (float*)send[i], count, (float)(i + 1));
    }
    if (!a.in_place)       // the check must see what the collective wrote, not what the buffer held before
      for (int i = 0; i < n; ++i) {
        HIPCHECK(hipSetDevice(i));
        HIPCHECK(hipMemsetAsync(recv[i], 0, bytes, streams[i]));
      }
    sync_all();
    const char* check = "skipped";
    round(count);
    sync_all();
    unsigned long long mismatched = 0;
    if (a.check) {
      const float want = (float)(n * (n + 1) / 2);
      const int grid = (int)std::min<size_t>(2048, (count + 255) / 256);
      for (int i = 0; i < n; ++i) {
        HIPCHECK(hipSetDevice(i));
        HIPCHECK(hipMemsetAsync(bad[i], 0, sizeof(unsigned long long), streams[i]));
        if (is_bf16) hipLaunchKernelGGL(count_mismatch<uint16_t>, dim3(grid), dim3(256), 0, streams[i], (const uint16_t*)recv[i], count, bf16_bits(want), bad[i]);
        else hipLaunchKernelGGL(count_mismatch<float>, dim3(grid), dim3(256), 0, streams[i], (const float*)recv[i], count, want, bad[i]);
      }
      for (int i = 0; i < n; ++i) {
        HIPCHECK(hipSetDevice(i));
        unsigned long long b = 0;
        HIPCHECK(hipMemcpyAsync(&b, bad[i], sizeof(b), hipMemcpyDeviceToHost, streams[i]));
        HIPCHECK(hipStreamSynchronize(streams[i]));
        mismatched += b;
      }
      check = mismatched == 0 ? "ok" : "MISMATCH";
    }
    for (int w = 0; w < a.warmup; ++w) round(count);
    sync_all();
    for (int i = 0; i < n; ++i) {
      HIPCHECK(hipSetDevice(i));
      HIPCHECK(hipEventRecord(ev0[i], streams[i]));
    }
    for (int it = 0; it < a.iters; ++it) round(count);
    for (int i = 0; i < n; ++i) {
      HIPCHECK(hipSetDevice(i));
      HIPCHECK(hipEventRecord(ev1[i], streams[i]));
    }
    sync_all();
    float ms = 0, ms_rank0 = 0;
    for (int i = 0; i < n; ++i) {
      HIPCHECK(hipSetDevice(i));
      HIPCHECK(hipEventSynchronize(ev1[i]));
      float m = 0;
      HIPCHECK(hipEventElapsedTime(&m, ev0[i], ev1[i]));
      if (i == 0) ms_rank0 = m;
      ms = std::max(ms, m);
    }
    const Point p = make_point(bytes, n, a.iters, ms, ms_rank0);
    if ((n > 1 ? p.bus : p.alg) > (n > 1 ? peak_bus : peak_alg)) peak_bytes = bytes;
    peak_bus = std::max(peak_bus, p.bus);
    peak_alg = std::max(peak_alg, p.alg);
    print_point(a, n, bytes, p, check, a.check ? count * (size_t)n : (size_t)0, mismatched);
  }
  print_summary(a, n, peak_bus, peak_alg, peak_bytes);

  for (int i = 0; i < n; ++i) {
    HIPCHECK(hipSetDevice(i));
    HIPCHECK(hipFree(send[i]));
    if (!a.in_place) HIPCHECK(hipFree(recv[i]));
    HIPCHECK(hipFree(bad[i]));
    HIPCHECK(hipEventDestroy(ev0[i]));
    HIPCHECK(hipEventDestroy(ev1[i]));
    HIPCHECK(hipStreamDestroy(streams[i]));
    ncclCommDestroy(comms[i]);
  }
  return 0;
}
