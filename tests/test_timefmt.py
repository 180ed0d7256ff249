"""parseDuration (monitor_server.js:54-63), pod age (:106-110), HH:MM labels (:141), toFixed(1)."""
import time

import pytest

from k8s_llm_monitor_amd.utils.timefmt import fixed1, format_age, hhmm, parse_duration, parse_k8s_time


@pytest.mark.parametrize("s,want", [("30m", 1800), ("30s", 30), ("2h", 7200), ("1d", 86400), ("15", 1800),
                                    ("", 1800), ("abc", 1800), ("5m30s", 300), ("x10sy", 10), (None, 1800), (30, 1800)])
def test_parse_duration(s, want):
    assert parse_duration(s) == want


@pytest.mark.parametrize("age,want", [(0, "0m"), (59, "0m"), (60, "1m"), (3599, "59m"), (3600, "1h 0m"),
                                      (3 * 3600 + 36 * 60 + 5, "3h 36m"), (86400, "1d 0h"), (2 * 86400 + 5 * 3600 + 59 * 60, "2d 5h")])
def test_format_age(age, want):
    assert format_age(1_000_000.0 - age, now=1_000_000.0) == want


def test_format_age_without_start_time_and_future_start():
    assert format_age(None) == "0m"                       # :105
    assert format_age(2000.0, now=1000.0) == "0m"         # clock skew never prints a negative age


def test_parse_k8s_time():
    assert parse_k8s_time("1970-01-01T00:01:00Z") == 60.0
    assert parse_k8s_time("1970-01-01T00:01:00.123456789Z") == pytest.approx(60.123456)
    assert parse_k8s_time("1970-01-01T01:00:00+01:00") == 0.0
    assert parse_k8s_time("garbage") is None and parse_k8s_time(None) is None and parse_k8s_time("") is None


def test_hhmm_is_local_24h():
    t = time.mktime((2026, 3, 13, 16, 5, 0, 0, 0, -1))
    assert hhmm(t) == "16:05"


@pytest.mark.parametrize("x,want", [(5.34, "5.3"), (5.35, "5.3"), (5.36, "5.4"), (0.25, "0.3"), (0.0, "0.0"),
                                    (100.0, "100.0"), (45.25, "45.3"), (-0.04, "0.0"), (1e-9, "0.0"), (99.95, "100.0")])
def test_fixed1_matches_js_tofixed(x, want):
    # 5.35 is 5.3499999... in binary: JS prints 5.3; exact ties (0.25, 45.25) round up in JS
    assert fixed1(x) == want


def test_fixed1_non_finite():
    assert fixed1(float("nan")) == "NaN" and fixed1(float("inf")) == "Infinity"
