"""Webhook notifier: alert begin / end events POSTed as JSON, off the sampling thread.

The reference has no notification channel: an alert exists only as the answer to a poll
(SURVEY.md 2.7 item 6). Events are queued by the AlertBook hook and sent by one worker
thread, so a slow or dead receiver can neither stall sampling nor lose ordering; the queue is
bounded and drops the oldest event when full.
"""
from __future__ import annotations

import json
import queue
import socket
import threading
import time
import urllib.request
from typing import Any, Dict, Optional

TIER_RANK = {"minor": 0, "serious": 1, "critical": 2}


class WebhookNotifier:
    def __init__(self, url: str, timeout: float = 3.0, min_tier: str = "minor", node: Optional[str] = None, max_queue: int = 256):
        self.url, self.timeout, self.min_rank = url, timeout, TIER_RANK[min_tier]
        self.node = node or socket.gethostname()
        self._q: "queue.Queue[Optional[Dict[str, Any]]]" = queue.Queue(maxsize=max_queue)
        self.sent = 0
        self.failed = 0
        self.dropped = 0
        self.last_error: Optional[str] = None
        self._th = threading.Thread(target=self._run, name="klm-webhook", daemon=True)
        self._th.start()

    def __call__(self, ev: Dict[str, Any]) -> None:
        tier = ev.get("tier") or "minor"
        rank = max(TIER_RANK.get(tier, 0), TIER_RANK.get(ev.get("previous_tier"), -1))   # a de-escalation is sent too
        if rank < self.min_rank:
            return
        doc = dict(ev, tier=tier, node=self.node, source="k8s_llm_monitor_amd")
        try:
            self._q.put_nowait(doc)
        except queue.Full:
            try:
                self._q.get_nowait()
                self.dropped += 1
                self._q.put_nowait(doc)
            except (queue.Empty, queue.Full):
                pass

    def _run(self) -> None:
        while True:
            doc = self._q.get()
            if doc is None:
                return
            try:
                req = urllib.request.Request(self.url, data=json.dumps(doc, ensure_ascii=False).encode("utf-8"),
                                             headers={"Content-Type": "application/json; charset=utf-8"}, method="POST")
                with urllib.request.urlopen(req, timeout=self.timeout) as r:
                    r.read()
                self.sent += 1
            except Exception as e:
                self.failed += 1
                self.last_error = f"{type(e).__name__}: {e}"

    def flush(self, timeout: float = 5.0) -> None:
        t0 = time.time()
        while not self._q.empty() and time.time() - t0 < timeout:
            time.sleep(0.01)
        time.sleep(0.02)

    def close(self) -> None:
        self._q.put(None)
        self._th.join(timeout=self.timeout + 1)

    def stats(self) -> Dict[str, Any]:
        return {"url": self.url, "sent": self.sent, "failed": self.failed, "dropped": self.dropped, "last_error": self.last_error}
