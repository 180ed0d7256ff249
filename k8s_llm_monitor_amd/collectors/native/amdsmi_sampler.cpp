// klm_amdsmi_sampler: native MI355X telemetry sampler on libamd_smi.
//
// No reference counterpart: the reference forks `nvidia-smi` per request
// (monitor_server.js:85). SURVEY.md 2.4 N1: per GPU HBM used/total, GFX and memory (UMC)
// activity, socket power, junction temperature, per-link xGMI counters, ECC counts and
// throttle residency, with no GPU work submitted.
//
// It polls amdsmi at `--hz` and prints one JSON line per `--period` seconds:
//   {"t": monotonic_s, "wall": epoch_s, "polls": n, "poll_us_mean": .., "gpus": [ {raw reading}, ... ]}
// A raw reading uses the key names of collectors/gpu.py (metrics_to_raw), so the Python
// side feeds it to the same GpuNormalizer as the in-process reader. Gauges are the mean of
// the polls of the period, `peak` holds their maxima (a 5 s dashboard period would
// otherwise miss a 200 ms power or temperature spike); accumulators are the latest value.
//
// amdsmi marks an unsupported field with an all-ones value of its width; those are
// printed as null.
#include <amd_smi/amdsmi.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

// the hash of the sources this binary was built from (build.py passes KLM_SRC_HASH and finds the marker to decide staleness)
#ifndef KLM_SRC_HASH
#define KLM_SRC_HASH "unhashed"
#endif
extern "C" __attribute__((used)) const char klm_src_hash_marker[] = "KLM_SRC_HASH=" KLM_SRC_HASH;

static volatile sig_atomic_t g_stop = 0;
static void on_signal(int) { g_stop = 1; }

static double mono_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}
static double wall_s() {
  using namespace std::chrono;
  return duration<double>(system_clock::now().time_since_epoch()).count();
}

struct Json {
  std::string s;
  bool first = true;
  void sep() {
    if (!first) s += ',';
    first = false;
  }
  void key(const char* k) {
    sep();
    s += '"';
    s += k;
    s += "\":";
  }
  void num(const char* k, double v) {
    key(k);
    if (std::isfinite(v)) {
      char b[40];
      snprintf(b, sizeof b, "%.10g", v);
      s += b;
    } else {
      s += "null";
    }
  }
  void u64(const char* k, uint64_t v) {
    key(k);
    s += std::to_string(v);
  }
  void null(const char* k) {
    key(k);
    s += "null";
  }
  void str(const char* k, const char* v) {
    key(k);
    s += '"';
    for (const char* p = v; *p; ++p) {
      if (*p == '"' || *p == '\\') s += '\\';
      if ((unsigned char)*p >= 0x20) s += *p;
    }
    s += '"';
  }
  template <class T>
  void opt(const char* k, T v) {  // all-ones = unsupported
    if (v == (T)~(T)0) null(k);
    else u64(k, (uint64_t)v);
  }
};

struct Gauge {
  double sum = 0, peak = -1e300;
  int n = 0;
  void add(double v) {
    sum += v;
    peak = std::max(peak, v);
    ++n;
  }
  double mean() const { return n ? sum / n : NAN; }
  double max() const { return n ? peak : NAN; }
  void reset() { *this = Gauge(); }
};

// What the topology calls said about one xGMI link slot at start-up (slot order = xgmi_read_data_acc order).
struct LinkTopo {
  std::string peer_bdf;      // "" = the far end is not a GPU this process can see
  int peer_index = -1;
  const char* source = nullptr;   // "link_metrics" | "xgmi_node_id"
  uint32_t max_gbps = 0;     // Gb/s, one direction
};

struct Gpu {
  amdsmi_processor_handle h;
  std::string name, uuid, bdf;
  std::vector<LinkTopo> links;
  bool have_xgmi_info = false;
  uint64_t hive_id = 0, node_id = 0;
  std::string topo_note, xgmi_err = "unknown";
  bool xgmi_err_supported = true;
  std::string compute_partition, memory_partition;     // "SPX" / "NPS1" ...; empty: the driver does not say
  bool partition_supported = true;
  bool have_bad_pages = false, bad_pages_supported = true;     // retired HBM pages (RAS), by status
  uint32_t bad_retired = 0, bad_pending = 0, bad_unreservable = 0;
  double cap_w = NAN, energy_res_uj = NAN;
  Gauge gfx, umc, power, temp, temp_mem, temp_edge, clk;
  amdsmi_gpu_metrics_t m;
  bool have_m = false;
  amdsmi_vram_usage_t vram{};
  bool have_vram = false;
  amdsmi_error_count_t ecc{};
  bool have_ecc = false;
  // --processes: the per-process VRAM table (slow call, off by default like the Python reader's)
  struct Proc { uint32_t pid; std::string name; uint64_t vram; };
  std::vector<Proc> procs;
  bool have_procs = false;
};

template <class T>
static bool ok(T v) { return v != (T)~(T)0; }

// One output line from the current per-GPU state; resets the period's gauges.
static std::string render_line(std::vector<Gpu>& gpus, double now, long polls, double poll_us) {
  Json j;
  j.s = "{";
  j.num("t", now);
  j.num("wall", wall_s());
  j.u64("polls", (uint64_t)polls);
  j.num("poll_us_mean", polls ? poll_us / polls : NAN);
  j.key("gpus");
  j.s += '[';
  for (size_t i = 0; i < gpus.size(); ++i) {
    Gpu& g = gpus[i];
    if (i) j.s += ',';
    Json r;
    r.s = "{";
    r.u64("index", i);
    r.num("t", now);
    r.num("wall", wall_s());
    r.str("name", g.name.c_str());
    r.str("uuid", g.uuid.c_str());
    r.str("bdf", g.bdf.c_str());
    r.num("power_cap_w", g.cap_w);
    r.num("energy_resolution_uj", g.energy_res_uj);
    if (g.have_vram) {
      r.u64("vram_used_mib", g.vram.vram_used);
      r.u64("vram_total_mib", g.vram.vram_total);
    }
    if (g.have_ecc) {
      r.u64("ecc_correctable", g.ecc.correctable_count);
      r.u64("ecc_uncorrectable", g.ecc.uncorrectable_count);
      r.u64("ecc_deferred", g.ecc.deferred_count);
    }
    if (!g.links.empty()) {
      auto arr = [&](const char* k, auto&& item) {
        r.key(k);
        r.s += '[';
        for (size_t l = 0; l < g.links.size(); ++l) {
          if (l) r.s += ',';
          item(g.links[l]);
        }
        r.s += ']';
      };
      auto quoted = [&](const std::string& v) { r.s += v.empty() ? std::string("null") : "\"" + v + "\""; };
      arr("xgmi_peers", [&](const LinkTopo& l) { quoted(l.peer_bdf); });
      arr("xgmi_peer_index", [&](const LinkTopo& l) { r.s += l.peer_index < 0 ? std::string("null") : std::to_string(l.peer_index); });
      arr("xgmi_peer_source", [&](const LinkTopo& l) { quoted(l.source ? l.source : ""); });
      arr("xgmi_link_max_gbps", [&](const LinkTopo& l) { r.s += std::to_string(l.max_gbps); });
    }
    if (g.have_xgmi_info) {
      r.u64("xgmi_hive_id", g.hive_id);
      r.u64("xgmi_node_id", g.node_id);
    }
    if (g.topo_note.empty()) r.null("xgmi_topology_note");
    else r.str("xgmi_topology_note", g.topo_note.c_str());
    r.str("xgmi_error_status", g.xgmi_err.c_str());
    if (g.have_bad_pages) {
      r.u64("bad_pages_retired", g.bad_retired);
      r.u64("bad_pages_pending", g.bad_pending);
      r.u64("bad_pages_unreservable", g.bad_unreservable);
    }
    if (!g.compute_partition.empty()) r.str("compute_partition", g.compute_partition.c_str());
    if (!g.memory_partition.empty()) r.str("memory_partition", g.memory_partition.c_str());
    if (g.have_m) {
      const amdsmi_gpu_metrics_t& m = g.m;
      r.num("gfx_activity", g.gfx.mean());
      r.num("umc_activity", g.umc.mean());
      r.num("power_w", g.power.mean());
      r.num("temp_hotspot", g.temp.mean());
      r.num("temp_mem", g.temp_mem.mean());
      r.num("clk_gfx", g.clk.mean());
      r.opt("clk_mem", m.current_uclk);
      r.opt("temp_edge", m.temperature_edge);
      r.opt("gfx_activity_acc", m.gfx_activity_acc);
      r.opt("mem_activity_acc", m.mem_activity_acc);
      r.opt("accumulation_counter", m.accumulation_counter);
      r.opt("firmware_timestamp", m.firmware_timestamp);
      r.opt("energy_acc", m.energy_accumulator);
      r.opt("res_prochot_acc", m.prochot_residency_acc);
      r.opt("res_ppt_acc", m.ppt_residency_acc);
      r.opt("res_socket_thm_acc", m.socket_thm_residency_acc);
      r.opt("res_vr_thm_acc", m.vr_thm_residency_acc);
      r.opt("res_hbm_thm_acc", m.hbm_thm_residency_acc);
      r.opt("xgmi_link_width", m.xgmi_link_width);
      r.opt("xgmi_link_speed", m.xgmi_link_speed);
      r.opt("pcie_bandwidth_inst", m.pcie_bandwidth_inst);
      // PCIe link health and shape (collectors/gpu.py metrics_to_raw: the replay alert and the link card read these)
      r.opt("pcie_bandwidth_acc", m.pcie_bandwidth_acc);
      r.opt("pcie_link_speed", m.pcie_link_speed);
      r.opt("pcie_link_width", m.pcie_link_width);
      r.opt("pcie_replay_acc", m.pcie_replay_count_acc);
      r.opt("pcie_replay_rollover_acc", m.pcie_replay_rover_count_acc);
      r.opt("pcie_l0_recovery_acc", m.pcie_l0_to_recov_count_acc);
      r.opt("pcie_nak_sent_acc", m.pcie_nak_sent_count_acc);
      r.opt("pcie_nak_rcvd_acc", m.pcie_nak_rcvd_count_acc);
      // per-XCD busy accumulators and gauges: the populated entries of every compute partition's row, in order (one row of
      // 8 in SPX mode, 8 rows of 1 in CPX) -- `_flatten_ints` on the Python side
      {
        std::string acc = "[", inst = "[";
        bool any_acc = false, any_inst = false;
        for (int p = 0; p < AMDSMI_MAX_NUM_XCP; ++p)
          for (int x = 0; x < AMDSMI_MAX_NUM_XCC; ++x) {
            if (ok(m.xcp_stats[p].gfx_busy_acc[x])) {
              if (any_acc) acc += ',';
              acc += std::to_string(m.xcp_stats[p].gfx_busy_acc[x]);
              any_acc = true;
            }
            if (ok(m.xcp_stats[p].gfx_busy_inst[x])) {
              if (any_inst) inst += ',';
              inst += std::to_string(m.xcp_stats[p].gfx_busy_inst[x]);
              any_inst = true;
            }
          }
        if (any_acc) { r.key("xcd_busy_acc"); r.s += acc + "]"; }
        if (any_inst) { r.key("xcd_busy_inst"); r.s += inst + "]"; }
      }
      auto arr64 = [&](const char* k, const uint64_t* a) {
        r.key(k);
        r.s += '[';
        for (int l = 0; l < AMDSMI_MAX_NUM_XGMI_LINKS; ++l) {
          if (l) r.s += ',';
          r.s += ok(a[l]) ? std::to_string(a[l]) : std::string("null");
        }
        r.s += ']';
      };
      arr64("xgmi_read_kib", m.xgmi_read_data_acc);
      arr64("xgmi_write_kib", m.xgmi_write_data_acc);
      r.key("xgmi_link_status");
      r.s += '[';
      for (int l = 0; l < AMDSMI_MAX_NUM_XGMI_LINKS; ++l) {
        if (l) r.s += ',';
        r.s += ok(m.xgmi_link_status[l]) ? std::to_string(m.xgmi_link_status[l]) : std::string("null");
      }
      r.s += ']';
      r.key("peak");
      Json p;
      p.s = "{";
      p.num("gfx_activity", g.gfx.max());
      p.num("umc_activity", g.umc.max());
      p.num("power_w", g.power.max());
      p.num("temp_hotspot", g.temp.max());
      p.num("temp_mem", g.temp_mem.max());
      p.num("temp_edge", g.temp_edge.max());
      p.s += '}';
      r.s += p.s;
    }
    if (g.have_procs) {
      r.u64("processes", g.procs.size());
      r.key("process_list");
      r.s += '[';
      for (size_t k = 0; k < g.procs.size(); ++k) {
        if (k) r.s += ',';
        Json q;
        q.s = "{";
        q.u64("pid", g.procs[k].pid);
        q.str("name", g.procs[k].name.c_str());
        q.u64("vramBytes", g.procs[k].vram);
        q.s += '}';
        r.s += q.s;
      }
      r.s += ']';
    }
    r.s += '}';
    j.s += r.s;
    g.gfx.reset();
    g.umc.reset();
    g.power.reset();
    g.temp.reset();
    g.temp_mem.reset();
    g.temp_edge.reset();
    g.clk.reset();
  }
  j.s += "]}";
  return j.s;
}

static std::string bdf_str(const amdsmi_bdf_t& bdf) {
  char b[40];
  snprintf(b, sizeof b, "%04llx:%02x:%02x.%x", (unsigned long long)bdf.domain_number, (unsigned)bdf.bus_number,
           (unsigned)bdf.device_number, (unsigned)bdf.function_number);
  return b;
}

// Name each link slot's far end: by the slot's PCI address when that is one of the GPUs we enumerate, else (same
// hive) by xgmi node id == slot number, else leave it unnamed and say why (collectors/gpu.py read_xgmi_topology
// is the same resolution for the in-process reader).
static void resolve_peers(std::vector<Gpu>& gpus) {
  for (size_t i = 0; i < gpus.size(); ++i) {
    Gpu& g = gpus[i];
    bool any = false;
    for (auto& l : g.links) {
      if (l.peer_bdf.empty()) continue;
      for (size_t j = 0; j < gpus.size(); ++j)
        if (j != i && gpus[j].bdf == l.peer_bdf) {
          l.peer_index = (int)j;
          l.source = "link_metrics";
          any = true;
        }
      if (l.peer_index < 0) l.peer_bdf.clear();
    }
    if (!any && gpus.size() > 1 && g.have_xgmi_info) {
      for (size_t slot = 0; slot < g.links.size(); ++slot)
        for (size_t j = 0; j < gpus.size(); ++j)
          if (j != i && gpus[j].have_xgmi_info && gpus[j].hive_id == g.hive_id && gpus[j].node_id == slot && g.node_id != slot) {
            g.links[slot].peer_index = (int)j;
            g.links[slot].peer_bdf = gpus[j].bdf;
            g.links[slot].source = "xgmi_node_id";
            any = true;
          }
    }
    if (!g.links.empty() && !any) {
      if (!g.topo_note.empty()) g.topo_note += "; ";
      g.topo_note += gpus.size() == 1 ? "peers not visible to this process (1 GPU visible of the hive)" : "no link slot names a visible GPU";
    }
  }
}

static void read_topology(Gpu& g) {
  amdsmi_xgmi_info_t xi;
  if (amdsmi_get_xgmi_info(g.h, &xi) == AMDSMI_STATUS_SUCCESS) {
    g.have_xgmi_info = true;
    g.hive_id = xi.xgmi_hive_id;
    g.node_id = xi.xgmi_node_id;
  } else {
    g.topo_note = "xgmi_info: unavailable";
  }
  static amdsmi_link_metrics_t lm;   // 5 KiB: keep it off the stack
  memset(&lm, 0, sizeof lm);
  if (amdsmi_get_link_metrics(g.h, &lm) == AMDSMI_STATUS_SUCCESS) {
    // MI355X / amdsmi 26.2 fills 8 slots and leaves num_links at 0 (profiles/r2_amdsmi_topo_mi355x.json)
    const uint32_t n = std::min<uint32_t>(std::max<uint32_t>(lm.num_links, AMDSMI_MAX_NUM_XGMI_LINKS), AMDSMI_MAX_NUM_XGMI_LINKS);
    for (uint32_t l = 0; l < n; ++l) {
      LinkTopo t;
      const amdsmi_bdf_t& b = lm.links[l].bdf;
      const bool unseen = b.bus_number == 0xff && b.device_number == 0x1f;   // all-ones address: not one of our devices
      if (!unseen) t.peer_bdf = bdf_str(b);
      t.max_gbps = lm.links[l].max_bandwidth;
      g.links.push_back(t);
    }
  } else {
    if (!g.topo_note.empty()) g.topo_note += "; ";
    g.topo_note += "link_metrics: unavailable";
  }
}

static void read_xgmi_error(Gpu& g) {
  if (!g.xgmi_err_supported) return;
  amdsmi_xgmi_status_t st;
  const amdsmi_status_t rc = amdsmi_gpu_xgmi_error_status(g.h, &st);
  if (rc == AMDSMI_STATUS_SUCCESS) {
    g.xgmi_err = st == AMDSMI_XGMI_STATUS_NO_ERRORS ? "no_errors" : st == AMDSMI_XGMI_STATUS_ERROR ? "error" : "multiple_errors";
  } else {
    g.xgmi_err_supported = false;      // MI355X answers NOT_SUPPORTED: ask once
    g.xgmi_err = rc == AMDSMI_STATUS_NOT_SUPPORTED ? "not_supported" : "error: call failed";
  }
}

// retired HBM pages: the count first (a null buffer asks for it), then at most 1024 records, counted by status
static void read_bad_pages(Gpu& g) {
  if (!g.bad_pages_supported) return;
  uint32_t n = 0;
  if (amdsmi_get_gpu_bad_page_info(g.h, &n, nullptr) != AMDSMI_STATUS_SUCCESS) {
    g.bad_pages_supported = false;
    return;
  }
  g.bad_retired = g.bad_pending = g.bad_unreservable = 0;
  if (n) {
    static amdsmi_retired_page_record_t recs[1024];
    uint32_t m = n < 1024 ? n : 1024;
    if (amdsmi_get_gpu_bad_page_info(g.h, &m, recs) != AMDSMI_STATUS_SUCCESS) return;      // keep the previous answer
    for (uint32_t k = 0; k < (m < 1024 ? m : 1024); ++k) {
      if (recs[k].status == AMDSMI_MEM_PAGE_STATUS_PENDING) ++g.bad_pending;
      else if (recs[k].status == AMDSMI_MEM_PAGE_STATUS_UNRESERVABLE) ++g.bad_unreservable;
      else ++g.bad_retired;
    }
    if (n > 1024) g.bad_retired += n - 1024;          // beyond the buffer: counted as retired
  }
  g.have_bad_pages = true;
}

// compute / memory partition modes: strings from the driver ("SPX", "NPS1"; "N/A" where there is none). Asked once where unsupported.
static void read_partitions(Gpu& g) {
  if (!g.partition_supported) return;
  char c[64] = {0}, m[64] = {0};
  const bool okc = amdsmi_get_gpu_compute_partition(g.h, c, sizeof c) == AMDSMI_STATUS_SUCCESS;
  const bool okm = amdsmi_get_gpu_memory_partition(g.h, m, sizeof m) == AMDSMI_STATUS_SUCCESS;
  c[sizeof c - 1] = m[sizeof m - 1] = 0;
  auto clean = [](const char* v) { return (!v[0] || !strcmp(v, "N/A") || !strcmp(v, "UNKNOWN")) ? std::string() : std::string(v); };
  g.compute_partition = okc ? clean(c) : std::string();
  g.memory_partition = okm ? clean(m) : std::string();
  if (!okc && !okm) g.partition_supported = false;
}

// --selftest: render one line from synthetic metrics (no driver needed), including
// unsupported-field sentinels and a 32-bit activity accumulator near its wrap. Lets the JSON
// writer run under host sanitizers and feeds the Python reader in CPU tests.
static int selftest() {
  std::vector<Gpu> gpus(2);
  for (size_t i = 0; i < gpus.size(); ++i) {
    Gpu& g = gpus[i];
    g.h = nullptr;
    g.name = i ? "AMD Instinct \"MI355X\"" : "AMD Instinct MI355X";
    g.uuid = "selftest-" + std::to_string(i);
    g.bdf = i ? "0000:75:00.0" : "0000:72:00.0";
    g.cap_w = 1400.0;
    g.energy_res_uj = 15.3;
    memset(&g.m, 0xFF, sizeof g.m);                 // every field "unsupported" ...
    g.have_m = true;
    g.m.gfx_activity_acc = 0xFFFFFFF0u + (uint32_t)i;   // ... then the ones MI355X reports
    g.m.mem_activity_acc = 1000;
    g.m.accumulation_counter = 23735538 + i;
    g.m.firmware_timestamp = 2399753411260ull;
    g.m.energy_accumulator = 456357518510ull;
    g.m.current_uclk = 2000;
    g.m.prochot_residency_acc = 0;
    g.m.ppt_residency_acc = 36266;
    g.m.socket_thm_residency_acc = 0;
    g.m.vr_thm_residency_acc = 0;
    g.m.hbm_thm_residency_acc = 0;
    g.m.xgmi_link_width = 16;
    g.m.xgmi_link_speed = 38;
    g.m.pcie_bandwidth_inst = 18;
    g.m.pcie_bandwidth_acc = 123456789ull;
    g.m.pcie_link_speed = 320;                        // 0.1 GT/s: Gen5
    g.m.pcie_link_width = 16;
    g.m.pcie_replay_count_acc = 7 + i;
    g.m.pcie_replay_rover_count_acc = 0;
    g.m.pcie_l0_to_recov_count_acc = 2;
    g.m.pcie_nak_sent_count_acc = 0;                  // pcie_nak_rcvd_count_acc stays all-ones: unsupported -> null
    for (int x = 0; x < AMDSMI_MAX_NUM_XCC; ++x) {    // SPX: partition 0 carries all eight XCDs, the other rows stay unsupported
      g.m.xcp_stats[0].gfx_busy_acc[x] = 1000000ull * (x + 1) + i;
      g.m.xcp_stats[0].gfx_busy_inst[x] = 10 * x;
    }
    if (i == 1) {
      g.procs.push_back({4242u, "python3 \"vllm\"", 17179869184ull});
      g.procs.push_back({4243u, "klm_burner", 2147483648ull});
      g.have_procs = true;
    }
    for (int l = 0; l < AMDSMI_MAX_NUM_XGMI_LINKS; ++l) {
      const bool self = (size_t)l == i;
      g.m.xgmi_read_data_acc[l] = self ? 0 : 43353304ull + l;
      g.m.xgmi_write_data_acc[l] = self ? 0 : 43520298ull + l;
      g.m.xgmi_link_status[l] = self ? 0xFFFF : 1;
    }
    g.have_xgmi_info = true;
    g.hive_id = 6319504153631384352ull;
    g.node_id = i;
    g.xgmi_err = "not_supported";
    if (i == 0) g.compute_partition = "SPX", g.memory_partition = "NPS1";      // GPU 1: the driver does not say
    if (i == 0) g.have_bad_pages = true, g.bad_retired = 2, g.bad_pending = 1;
    for (int l = 0; l < AMDSMI_MAX_NUM_XGMI_LINKS; ++l) {
      LinkTopo t;
      t.max_gbps = 608;
      if (l < 2 && (size_t)l != i) t.peer_bdf = l ? "0000:75:00.0" : "0000:72:00.0";   // the other selftest GPU; slots 2..7 unseen
      g.links.push_back(t);
    }
    for (double v : {10.0, 30.0}) g.gfx.add(v);
    g.umc.add(5);
    for (double v : {300.0, 900.0}) g.power.add(v);
    g.temp.add(46);
    g.temp_mem.add(34);
    g.clk.add(2100);
    g.vram.vram_total = 294896;
    g.vram.vram_used = 284 + 1000 * (uint32_t)i;
    g.have_vram = true;
    g.ecc.correctable_count = i;
    g.ecc.uncorrectable_count = 0;
    g.ecc.deferred_count = 0;
    g.have_ecc = true;
  }
  resolve_peers(gpus);
  puts(render_line(gpus, 1234.5, 20, 6000.0).c_str());
  return 0;
}

int main(int argc, char** argv) {
  double period = 1.0, hz = 20.0;
  bool once = false, processes = false;
  long max_lines = 0;
  for (int i = 1; i < argc; ++i) {
    std::string k = argv[i];
    if (k == "--period" && i + 1 < argc) period = atof(argv[++i]);
    else if (k == "--hz" && i + 1 < argc) hz = atof(argv[++i]);
    else if (k == "--lines" && i + 1 < argc) max_lines = atol(argv[++i]);
    else if (k == "--once") once = true;
    else if (k == "--processes") processes = true;
    else if (k == "--selftest") return selftest();
    else {
      fprintf(stderr, "usage: klm_amdsmi_sampler [--period S] [--hz H] [--lines N] [--once] [--processes] [--selftest]\n");
      return k == "--help" ? 0 : 1;
    }
  }
  if (period <= 0 || hz <= 0) {
    printf("{\"error\":\"period and hz must be positive\"}\n");
    return 1;
  }
  signal(SIGINT, on_signal);
  signal(SIGTERM, on_signal);
  signal(SIGPIPE, on_signal);

  amdsmi_status_t st = amdsmi_init(AMDSMI_INIT_AMD_GPUS);
  if (st != AMDSMI_STATUS_SUCCESS) {
    const char* msg = "?";
    amdsmi_status_code_to_string(st, &msg);
    printf("{\"error\":\"amdsmi_init: %s\"}\n", msg);
    return 2;
  }
  std::vector<Gpu> gpus;
  uint32_t ns = 0;
  amdsmi_get_socket_handles(&ns, nullptr);
  std::vector<amdsmi_socket_handle> socks(ns);
  amdsmi_get_socket_handles(&ns, socks.data());
  for (uint32_t s = 0; s < ns; ++s) {
    uint32_t np = 0;
    amdsmi_get_processor_handles(socks[s], &np, nullptr);
    std::vector<amdsmi_processor_handle> ph(np);
    amdsmi_get_processor_handles(socks[s], &np, ph.data());
    for (uint32_t p = 0; p < np; ++p) {
      processor_type_t ty;
      if (amdsmi_get_processor_type(ph[p], &ty) == AMDSMI_STATUS_SUCCESS && ty != AMDSMI_PROCESSOR_TYPE_AMD_GPU) continue;
      Gpu g;
      g.h = ph[p];
      amdsmi_asic_info_t asic;
      if (amdsmi_get_gpu_asic_info(g.h, &asic) == AMDSMI_STATUS_SUCCESS) g.name = asic.market_name;
      char uuid[AMDSMI_GPU_UUID_SIZE + 2] = {0};
      unsigned int ul = AMDSMI_GPU_UUID_SIZE;
      if (amdsmi_get_gpu_device_uuid(g.h, &ul, uuid) == AMDSMI_STATUS_SUCCESS) g.uuid = uuid;
      amdsmi_bdf_t bdf;
      if (amdsmi_get_gpu_device_bdf(g.h, &bdf) == AMDSMI_STATUS_SUCCESS) g.bdf = bdf_str(bdf);
      read_topology(g);
      amdsmi_power_cap_info_t cap;
      if (amdsmi_get_power_cap_info(g.h, 0, &cap) == AMDSMI_STATUS_SUCCESS && cap.power_cap)
        g.cap_w = cap.power_cap > 100000 ? cap.power_cap / 1e6 : (double)cap.power_cap;
      uint64_t e, ts;
      float res;
      if (amdsmi_get_energy_count(g.h, &e, &res, &ts) == AMDSMI_STATUS_SUCCESS) g.energy_res_uj = res;
      gpus.push_back(g);
    }
  }
  if (gpus.empty()) {
    printf("{\"error\":\"amdsmi found no GPU\"}\n");
    amdsmi_shut_down();
    return 3;
  }
  resolve_peers(gpus);

  const double dt_poll = 1.0 / hz;
  double next_emit = mono_s() + (once ? 0.0 : period);
  double next_poll = mono_s();
  long polls = 0, lines = 0;
  double poll_us = 0;
  int slow = 0;
  while (!g_stop) {
    const double t0 = mono_s();
    for (auto& g : gpus) {
      if (amdsmi_get_gpu_metrics_info(g.h, &g.m) == AMDSMI_STATUS_SUCCESS) {
        g.have_m = true;
        if (ok(g.m.average_gfx_activity)) g.gfx.add(g.m.average_gfx_activity);
        if (ok(g.m.average_umc_activity)) g.umc.add(g.m.average_umc_activity);
        if (ok(g.m.current_socket_power)) g.power.add(g.m.current_socket_power);
        else if (ok(g.m.average_socket_power)) g.power.add(g.m.average_socket_power);
        if (ok(g.m.temperature_hotspot)) g.temp.add(g.m.temperature_hotspot);
        if (ok(g.m.temperature_mem)) g.temp_mem.add(g.m.temperature_mem);
        if (ok(g.m.temperature_edge)) g.temp_edge.add(g.m.temperature_edge);
        if (ok(g.m.current_gfxclk)) g.clk.add(g.m.current_gfxclk);
      }
    }
    poll_us += (mono_s() - t0) * 1e6;
    ++polls;
    const double now = mono_s();
    if (now >= next_emit) {
      // VRAM and ECC once per line: they are separate driver calls and change slowly
      for (auto& g : gpus) {
        g.have_vram = amdsmi_get_gpu_vram_usage(g.h, &g.vram) == AMDSMI_STATUS_SUCCESS;
        if (slow % 10 == 0 || !g.have_ecc) {
          g.have_ecc = amdsmi_get_gpu_total_ecc_count(g.h, &g.ecc) == AMDSMI_STATUS_SUCCESS;
          read_xgmi_error(g);
          read_partitions(g);
          read_bad_pages(g);
          if (processes) {
            static amdsmi_proc_info_t list[64];
            uint32_t n = 64;
            const amdsmi_status_t rc = amdsmi_get_gpu_process_list(g.h, &n, list);
            g.have_procs = rc == AMDSMI_STATUS_SUCCESS || rc == AMDSMI_STATUS_OUT_OF_RESOURCES;
            g.procs.clear();
            if (g.have_procs)
              for (uint32_t k = 0; k < std::min<uint32_t>(n, 64); ++k) {
                list[k].name[AMDSMI_MAX_STRING_LENGTH - 1] = 0;
                g.procs.push_back({list[k].pid, list[k].name, list[k].memory_usage.vram_mem ? list[k].memory_usage.vram_mem : list[k].mem});
              }
          }
        }
      }
      ++slow;
      const std::string line = render_line(gpus, now, polls, poll_us);
      puts(line.c_str());
      fflush(stdout);
      if (ferror(stdout)) break;
      ++lines;
      polls = 0;
      poll_us = 0;
      if (once || (max_lines && lines >= max_lines)) break;
      next_emit += period;
      if (next_emit < now) next_emit = now + period;
    }
    next_poll += dt_poll;
    double wait = next_poll - mono_s();
    if (wait < 0) {
      next_poll = mono_s();
      wait = 0;
    }
    // sleep in short slices so a signal ends the loop promptly
    while (wait > 0 && !g_stop) {
      const double sl = std::min(wait, 0.05);
      std::this_thread::sleep_for(std::chrono::duration<double>(sl));
      wait -= sl;
    }
  }
  amdsmi_shut_down();
  return 0;
}
