#!/usr/bin/env python
"""Socket power + shader clock trace of GPU 0 at >= 10 Hz (measurement infrastructure, not product code).

Why: DESIGN §4 / §9 argue that the dense kernels of the contract step are POWER-bound (the chip holds its budget by lowering the clock on
random operands).  Until round 4 that rested on clocks inferred from GRBM_GUI_ACTIVE cycle counts; this puts a measured wattage / clock
trace beside it.

Two uses:
  * as a library: ``PowerSampler(hz=20).start()`` ... ``.stop()`` -> list of samples, ``summarize(samples, t0, t1)``;
    ``bench.py --power-trace`` does this around its timed region and adds a ``power`` object to its JSON line;
  * as a wrapper:  ``python scripts/power_trace.py --out gpurun_out/x/power -- python bench.py --steps 20``
    runs the command, samples for its whole life time, writes ``<out>.csv`` (every sample) and ``<out>.json`` (summary; when the command's
    stdout holds a bench JSON line with ``timed_region_unix``, the summary is cut to that window as well).

Sources, first one that works: the ``amdsmi`` Python binding (amdsmi_get_gpu_metrics_info: current / average socket power, current gfxclk;
amdsmi_get_power_info, amdsmi_get_clock_info), then the amdgpu hwmon / pp_dpm_sclk sysfs files.  Every sample records its source."""
from __future__ import annotations

import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time


def _num(v):
    try:
        f = float(v)
        return f if f == f and abs(f) < 1e9 else None
    except (TypeError, ValueError):
        return None


class _AmdSmi:
    name = "amdsmi"

    def __init__(self, index=0):
        import amdsmi
        self.m = amdsmi
        amdsmi.amdsmi_init()
        self.h = amdsmi.amdsmi_get_processor_handles()[index]
        self.read()  # raises if nothing readable

    def read(self):
        m, out = self.m, {}
        try:
            g = m.amdsmi_get_gpu_metrics_info(self.h)
            out["power_w"] = _num(g.get("current_socket_power")) or _num(g.get("average_socket_power"))
            clk = g.get("current_gfxclks") or g.get("current_gfxclk")
            if isinstance(clk, (list, tuple)):  # one entry per XCD on MI300-class parts
                vals = [c for c in (_num(v) for v in clk) if c and c < 60000]
                if vals:
                    out["sclk_mhz"], out["sclk_min_mhz"], out["sclk_max_mhz"] = sum(vals) / len(vals), min(vals), max(vals)
            elif _num(clk):
                out["sclk_mhz"] = _num(clk)
            out["temp_hotspot_c"] = _num(g.get("temperature_hotspot"))
            out["gfx_activity"] = _num(g.get("average_gfx_activity"))
            thr = g.get("throttle_status")
            out["throttle_status"] = thr if isinstance(thr, (int, float)) else None
        except Exception:  # noqa: BLE001
            pass
        if out.get("power_w") is None:
            p = m.amdsmi_get_power_info(self.h)
            out["power_w"] = _num(p.get("current_socket_power")) or _num(p.get("average_socket_power")) or _num(p.get("socket_power"))
        if out.get("sclk_mhz") is None:
            c = m.amdsmi_get_clock_info(self.h, m.AmdSmiClkType.GFX)
            out["sclk_mhz"] = _num(c.get("clk")) or _num(c.get("cur_clk"))
        if out.get("power_w") is None and out.get("sclk_mhz") is None:
            raise RuntimeError("amdsmi: neither power nor clock readable")
        return out


class _Sysfs:
    name = "sysfs"

    def __init__(self, index=0):
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        if not cards:
            raise RuntimeError("no amdgpu hwmon directory")
        self.hw = cards[min(index, len(cards) - 1)]
        self.dev = os.path.dirname(os.path.dirname(self.hw))
        self.read()

    @staticmethod
    def _cat(p):
        try:
            with open(p) as f:
                return f.read().strip()
        except OSError:
            return None

    def read(self):
        out = {}
        for f in ("power1_input", "power1_average"):
            v = _num(self._cat(os.path.join(self.hw, f)))
            if v is not None:
                out["power_w"] = v / 1e6
                break
        v = _num(self._cat(os.path.join(self.hw, "freq1_input")))
        if v is not None:
            out["sclk_mhz"] = v / 1e6
        else:
            txt = self._cat(os.path.join(self.dev, "pp_dpm_sclk")) or ""
            for line in txt.splitlines():
                if line.rstrip().endswith("*"):
                    out["sclk_mhz"] = _num(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
        if not out:
            raise RuntimeError("sysfs: neither power nor clock readable")
        return out


class PowerSampler:

    def __init__(self, hz: float = 20.0, index: int = 0):
        self.period = 1.0 / hz
        self.samples = []
        self.errors = []
        self.src = None
        for cls in (_AmdSmi, _Sysfs):
            try:
                self.src = cls(index)
                break
            except Exception as e:  # noqa: BLE001
                self.errors.append(f"{cls.name}: {e!r}"[:200])
        self._stop = threading.Event()
        self._thr = None

    @property
    def available(self):
        return self.src is not None

    def _loop(self):
        nxt = time.time()
        while not self._stop.is_set():
            try:
                s = self.src.read()
                s["t"] = time.time()
                self.samples.append(s)
            except Exception as e:  # noqa: BLE001
                if len(self.errors) < 8:
                    self.errors.append(repr(e)[:200])
            nxt += self.period
            d = nxt - time.time()
            if d > 0:
                self._stop.wait(d)
            else:
                nxt = time.time()

    def start(self):
        if self.src is not None:
            self._thr = threading.Thread(target=self._loop, daemon=True)
            self._thr.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thr is not None:
            self._thr.join(timeout=2.0)
        return self.samples


def _stats(vals):
    vals = sorted(v for v in vals if v is not None)
    if not vals:
        return None
    n = len(vals)
    return {"mean": round(sum(vals) / n, 2), "min": round(vals[0], 2), "p50": round(vals[n // 2], 2), "max": round(vals[-1], 2), "n": n}


def summarize(samples, t0=None, t1=None, source=None):
    win = [s for s in samples if (t0 is None or s["t"] >= t0) and (t1 is None or s["t"] <= t1)]
    out = {"source": source, "samples": len(win), "window_s": round((win[-1]["t"] - win[0]["t"]), 3) if len(win) > 1 else 0.0,
           "rate_hz": round((len(win) - 1) / (win[-1]["t"] - win[0]["t"]), 1) if len(win) > 1 and win[-1]["t"] > win[0]["t"] else None,
           "power_w": _stats([s.get("power_w") for s in win]), "sclk_mhz": _stats([s.get("sclk_mhz") for s in win]),
           "sclk_min_over_xcds_mhz": _stats([s.get("sclk_min_mhz") for s in win]),
           "temp_hotspot_c": _stats([s.get("temp_hotspot_c") for s in win])}
    thr = [s.get("throttle_status") for s in win if s.get("throttle_status") is not None]
    if thr:
        out["throttle_status_nonzero_share"] = round(sum(1 for v in thr if v) / len(thr), 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True, help="prefix: <out>.csv (samples) and <out>.json (summary)")
    ap.add_argument("--hz", type=float, default=20.0)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    if not cmd:
        raise SystemExit("usage: power_trace.py --out PREFIX -- <command>")
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    ps = PowerSampler(a.hz).start()
    t_start = time.time()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    t_end = time.time()
    samples = ps.stop()
    sys.stdout.write(r.stdout)
    keys = ["t", "power_w", "sclk_mhz", "sclk_min_mhz", "sclk_max_mhz", "temp_hotspot_c", "gfx_activity", "throttle_status"]
    with open(a.out + ".csv", "w") as f:
        f.write(",".join(keys) + "\n")
        for s in samples:
            f.write(",".join("" if s.get(k) is None else (f"{s[k]:.4f}" if k == "t" else str(s[k])) for k in keys) + "\n")
    src = ps.src.name if ps.available else None
    summ = {"command": " ".join(cmd), "returncode": r.returncode, "sampler_errors": ps.errors, "whole_run": summarize(samples, t_start, t_end, src)}
    for line in r.stdout.splitlines():
        if line.startswith("{") and "timed_region_unix" in line:
            try:
                j = json.loads(line)
                w0, w1 = j["timed_region_unix"]
                summ["timed_region"] = summarize(samples, w0, w1, src)
                summ["bench"] = {k: j.get(k) for k in ("ms_per_step", "value", "step_tflops")}
                summ["bench"]["attn_tflops"] = (j.get("roofline") or {}).get("achieved")
            except Exception as e:  # noqa: BLE001
                summ["timed_region_error"] = repr(e)[:200]
    with open(a.out + ".json", "w") as f:
        json.dump(summ, f, indent=1)
    print(json.dumps({"power_trace": summ.get("timed_region", summ["whole_run"])}), file=sys.stderr)
    sys.exit(r.returncode)


if __name__ == "__main__":
    main()
