"""Webhook notifications and debouncing (neither exists in the reference)."""
import json
import threading
from http.server import BaseHTTPRequestHandler, HTTPServer

import pytest

from k8s_llm_monitor_amd.collectors.gpu import GpuSource
from k8s_llm_monitor_amd.config import demo_config
from k8s_llm_monitor_amd.fakes.gpu import FakeGpuReader, GpuState
from k8s_llm_monitor_amd.fakes.kube import FakeKube
from k8s_llm_monitor_amd.server.alerts import AlertBook, empty_alerts
from k8s_llm_monitor_amd.server.sampler import Monitor


@pytest.fixture
def receiver():
    got = []

    class H(BaseHTTPRequestHandler):
        def log_message(self, *a):
            pass

        def do_POST(self):
            got.append(json.loads(self.rfile.read(int(self.headers["Content-Length"]))))
            self.send_response(204)
            self.end_headers()

    srv = HTTPServer(("127.0.0.1", 0), H)
    threading.Thread(target=srv.serve_forever, daemon=True).start()
    yield f"http://127.0.0.1:{srv.server_address[1]}/hook", got
    srv.shutdown()


def test_begin_and_end_events_reach_the_webhook(receiver, clock):
    url, got = receiver
    level = {"hbm": 0.5}
    cfg = demo_config(notify={"webhook_url": url, "min_tier": "serious"})
    mon = Monitor(cfg, gpu=GpuSource(FakeGpuReader(1, lambda t, i: GpuState(hbm_frac=level["hbm"]), clock=clock)),
                  kube=FakeKube("empty", clock=clock), clock=clock)
    mon.tick_all()
    level["hbm"] = 0.72          # minor: below min_tier, not sent
    clock.advance(5); mon.tick_all()
    level["hbm"] = 0.97          # escalates to critical: sent as a "change"
    clock.advance(5); mon.tick_all()
    level["hbm"] = 0.5
    clock.advance(5); mon.tick_all()
    mon.notifier.flush()
    st = mon.stats()["webhook"]
    mon.stop()
    ev = [(e["event"], e["key"], e["tier"]) for e in got]
    assert ev == [("change", "gpu/0/hbm", "critical"), ("end", "gpu/0/hbm", "critical")]
    assert got[0]["previous_tier"] == "minor"
    assert [h["event"] for h in mon.book.history] == ["begin", "change", "end"]
    begin = got[0]
    assert begin["source"] == "k8s_llm_monitor_amd" and "HBM" in begin["title"] and begin["fix"] and begin["node"]
    assert st["sent"] >= 2 and st["failed"] == 0


def test_dead_webhook_never_blocks_sampling(clock):
    cfg = demo_config(notify={"webhook_url": "http://127.0.0.1:9/hook", "webhook_timeout": 0.2})
    mon = Monitor(cfg, gpu=GpuSource(FakeGpuReader(1, "hbm_ramp", clock=clock)), kube=FakeKube("empty", clock=clock), clock=clock)
    import time
    t0 = time.time()
    for _ in range(5):
        clock.advance(10)
        mon.tick_all()
    assert time.time() - t0 < 1.0
    mon.notifier.flush(2.0)
    assert mon.notifier.failed >= 1 and mon.notifier.last_error
    mon.stop()


def test_debounce_hides_short_spikes():
    book = AlertBook(hold_s=30.0, debounce_s=10.0)
    spike = empty_alerts()
    spike["serious"].append({"title": "t", "desc": "d", "fix": "f", "key": "host/cpu", "tier": "serious"})
    assert book.update(spike, 0.0)["serious"] == []                # too young
    assert book.update(spike, 5.0)["serious"] == []
    assert book.update(empty_alerts(), 6.0)["serious"] == [] and book.history == []        # gone before it was ever shown
    assert book.update(spike, 20.0)["serious"] == []
    shown = book.update(spike, 31.0)["serious"]
    assert len(shown) == 1 and shown[0]["since"] == 20.0 and [h["event"] for h in book.history] == ["begin"]
    book.update(empty_alerts(), 40.0)
    assert [h["event"] for h in book.history] == ["begin", "end"]
