// klm_rccl_probe: RCCL all-reduce bandwidth probe over xGMI for MI355X nodes.
//
// No reference counterpart (the reference has no collective call site);
// BASELINE.json north_star item 4 / SURVEY.md 2.4 N5: "ncclAllReduce sum at 1, 2, 4, 8 GPUs
// over xGMI, algBW and busBW per size, JSON the server exposes and the dashboard plots".
//
// One process, one communicator per device (ncclCommInitAll), all ranks driven from this
// thread inside ncclGroupStart/End. Per message size: `warmup` untimed rounds, then `iters`
// rounds timed with hipEvents on rank 0's stream between two cross-device syncs; every
// stream is synchronised before the clock stops, so the time is the slowest rank's.
// Output: JSON lines on stdout (parallel/probe.py parses them):
//   {"type":"header",...}  {"type":"point",...}*  {"type":"summary",...}   or {"type":"error",...}
//
// busBW = algBW * 2(n-1)/n (nccl-tests convention). n = 1 has no link traffic: RCCL does a
// device-local copy, reported as algBW with busBW 0.
//
// Data check: each rank's buffer holds (rank + 1) before the first all-reduce of a size; the
// sum must be n(n+1)/2 in every element (exact in bf16 and fp32 for n <= 8).
#include <hip/hip_runtime.h>
#include <rccl.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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
  bool check = true;
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

int main(int argc, char** argv) {
  Args a;
  if (!parse(argc, argv, a)) {
    fprintf(stderr, "usage: klm_rccl_probe [--gpus N] [--dtype bf16|fp32] [--min B] [--max B] [--factor F] "
                    "[--warmup W] [--iters I] [--no-check]\n");
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
  int have = 0;
  HIPCHECK(hipGetDeviceCount(&have));
  if (have < a.gpus) {
    printf("{\"type\":\"error\",\"error\":\"need %d devices, have %d\"}\n", a.gpus, have);
    return 4;
  }
  const int n = a.gpus;
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) devs[i] = i;
  std::vector<ncclComm_t> comms(n);
  NCCLCHECK(ncclCommInitAll(comms.data(), n, devs.data()));
  int ver = 0;
  ncclGetVersion(&ver);

  std::vector<hipStream_t> streams(n);
  std::vector<void*> send(n), recv(n);
  for (int i = 0; i < n; ++i) {
    HIPCHECK(hipSetDevice(i));
    HIPCHECK(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));
    HIPCHECK(hipMalloc(&send[i], a.max_bytes));
    HIPCHECK(hipMalloc(&recv[i], a.max_bytes));
  }
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, 0));
  printf("{\"type\":\"header\",\"n_gpus\":%d,\"dtype\":\"%s\",\"op\":\"allreduce\",\"warmup\":%d,\"iters\":%d,"
         "\"rccl_version\":%d,\"device\":\"%s\",\"arch\":\"%s\",\"min_bytes\":%zu,\"max_bytes\":%zu,\"factor\":%d}\n",
         n, a.dtype.c_str(), a.warmup, a.iters, ver, prop.name, prop.gcnArchName, a.min_bytes, a.max_bytes, a.factor);
  fflush(stdout);

  HIPCHECK(hipSetDevice(0));
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));

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

  std::vector<size_t> sizes;
  for (size_t b = a.min_bytes; b < a.max_bytes; b *= (size_t)a.factor) sizes.push_back(b);
  sizes.push_back(a.max_bytes);

  double peak_bus = 0, peak_alg = 0;
  size_t peak_bytes = 0;
  const double bus_factor = n > 1 ? 2.0 * (n - 1) / n : 0.0;
  for (size_t bytes : sizes) {
    bytes -= bytes % esz;
    const size_t count = bytes / esz;
    for (int i = 0; i < n; ++i) {
      HIPCHECK(hipSetDevice(i));
      const int grid = (int)std::min<size_t>(2048, (count + 255) / 256);
      if (is_bf16) hipLaunchKernelGGL(fill16, dim3(grid), dim3(256), 0, streams[i], (uint16_t*)send[i], count, bf16_bits((float)(i + 1)));
      else hipLaunchKernelGGL(fill32, dim3(grid), dim3(256), 0, streams[i], (float*)send[i], count, (float)(i + 1));
    }
    sync_all();
    const char* check = "skipped";
    round(count);
    sync_all();
    if (a.check) {
      const double want = n * (n + 1) / 2.0;
      bool ok = true;
      const size_t probe_idx[3] = {0, count / 2, count - 1};
      for (int i = 0; i < n && ok; ++i) {
        HIPCHECK(hipSetDevice(i));
        for (size_t idx : probe_idx) {
          double got;
          if (is_bf16) {
            uint16_t h;
            HIPCHECK(hipMemcpy(&h, (uint16_t*)recv[i] + idx, 2, hipMemcpyDeviceToHost));
            got = bf16_to_float(h);
          } else {
            float f;
            HIPCHECK(hipMemcpy(&f, (float*)recv[i] + idx, 4, hipMemcpyDeviceToHost));
            got = f;
          }
          if (got != want) ok = false;
        }
      }
      check = ok ? "ok" : "MISMATCH";
    }
    for (int w = 0; w < a.warmup; ++w) round(count);
    sync_all();
    HIPCHECK(hipSetDevice(0));
    HIPCHECK(hipEventRecord(e0, streams[0]));
    for (int it = 0; it < a.iters; ++it) round(count);
    HIPCHECK(hipSetDevice(0));
    HIPCHECK(hipEventRecord(e1, streams[0]));
    sync_all();
    HIPCHECK(hipSetDevice(0));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    const double sec = (double)ms * 1e-3 / a.iters;
    const double alg = sec > 0 ? bytes / sec / 1e9 : 0.0;
    const double bus = alg * bus_factor;
    if ((n > 1 ? bus : alg) > (n > 1 ? peak_bus : peak_alg)) peak_bytes = bytes;
    peak_bus = std::max(peak_bus, bus);
    peak_alg = std::max(peak_alg, alg);
    printf("{\"type\":\"point\",\"n_gpus\":%d,\"bytes\":%zu,\"dtype\":\"%s\",\"op\":\"allreduce\",\"time_us\":%.3f,"
           "\"algbw_GBps\":%.4f,\"busbw_GBps\":%.4f,\"check\":\"%s\"}\n",
           n, bytes, a.dtype.c_str(), sec * 1e6, alg, bus, check);
    fflush(stdout);
  }
  printf("{\"type\":\"summary\",\"n_gpus\":%d,\"dtype\":\"%s\",\"peak_busbw_GBps\":%.4f,\"peak_algbw_GBps\":%.4f,"
         "\"peak_bytes\":%zu,\"ceiling_mesh_GBps\":%.1f,\"ceiling_ring_GBps\":%.1f}\n",
         n, a.dtype.c_str(), peak_bus, peak_alg, peak_bytes, n > 1 ? (n - 1) * 153.0 : 0.0, n > 1 ? 153.0 : 0.0);
  fflush(stdout);

  for (int i = 0; i < n; ++i) {
    HIPCHECK(hipSetDevice(i));
    HIPCHECK(hipFree(send[i]));
    HIPCHECK(hipFree(recv[i]));
    HIPCHECK(hipStreamDestroy(streams[i]));
    ncclCommDestroy(comms[i]);
  }
  return 0;
}
