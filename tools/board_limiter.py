"""Which limiter holds the shader clock while a command runs?  Samples the board's power-management read-outs through the
amdsmi Python package that ships with ROCm (/opt/rocm/share/amd_smi) twice a second next to a child command:

    python tools/board_limiter.py --out gpurun_out/limiter_conv.json -- python tools/conv_bench.py --reps 3000

Per sample: socket power and its limit, gfx clock (per XCD from gpu_metrics), hotspot / memory temperature, and the firmware's
VIOLATION ACCUMULATORS (amdsmi_get_violation_status: counters the power-management firmware increments every control iteration in which
a given controller held the clock down -- PPT = package power tracking, socket / VR / HBM thermal, PROCHOT, and since gpu_metrics 1.8
"gfx clock below host limit" split into power / thermal per XCD).  The read-out that names the limiter is the DELTA of an accumulator
over the run divided by the delta of `acc_counter` (= share of control iterations in which that controller was active).
Everything is best effort: a call the driver does not support is recorded as its exception text."""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, "/opt/rocm/share/amd_smi")


def _try(f, *a):
    try:
        return f(*a)
    except Exception as e:      # amdsmi raises its own exception types per status code
        return f"ERR {type(e).__name__}: {e}"


def _plain(x):
    if isinstance(x, dict):
        return {str(k): _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    if isinstance(x, (int, float, str, bool)) or x is None:
        return x
    return str(x)


class Limiter:
    """amdsmi handle of one GPU (by index in amdsmi's own enumeration, or by PCI bus/device/function prefix)."""

    def __init__(self, index=0, bdf=None):
        import amdsmi
        self.a = amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        self.h = hs[index]
        if bdf:
            for h in hs:
                b = _try(amdsmi.amdsmi_get_gpu_device_bdf, h)
                if isinstance(b, str) and b.lower().startswith(bdf.lower()):
                    self.h = h
                    break

    def violation(self):
        return _plain(_try(self.a.amdsmi_get_violation_status, self.h))

    def sample(self, full=False):
        a, h = self.a, self.h
        s = {"t": time.time(), "power": _plain(_try(a.amdsmi_get_power_info, h))}
        m = _try(a.amdsmi_get_gpu_metrics_info, h)
        if isinstance(m, dict):
            keep = ("temperature_hotspot", "temperature_mem", "temperature_vrsoc", "curr_socket_power", "average_socket_power", "average_gfx_activity",
                    "average_umc_activity", "throttle_status", "indep_throttle_status", "current_gfxclks", "current_gfxclk", "current_uclk",
                    "current_socclks", "gfxclk_lock_status", "accumulation_counter", "prochot_residency_acc", "ppt_residency_acc",
                    "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "energy_accumulator", "firmware_timestamp")
            s["metrics"] = _plain(m if full else {k: m[k] for k in keep if k in m})
        else:
            s["metrics"] = m
        s["violation"] = self.violation()
        return s

    def close(self):
        _try(self.a.amdsmi_shut_down)


def accumulator_shares(v0, v1):
    """Share of the firmware's control iterations between two violation read-outs in which each controller was active."""
    if not (isinstance(v0, dict) and isinstance(v1, dict)):
        return None
    n = (v1.get("acc_counter") or 0) - (v0.get("acc_counter") or 0) if isinstance(v1.get("acc_counter"), int) and isinstance(v0.get("acc_counter"), int) else 0
    out = {"control_iterations": n}
    for k in ("acc_prochot_thrm", "acc_ppt_pwr", "acc_socket_thrm", "acc_vr_thrm", "acc_hbm_thrm", "acc_gfx_clk_below_host_limit"):
        a, b = v0.get(k), v1.get(k)
        if isinstance(a, int) and isinstance(b, int):
            out[k.replace("acc_", "share_")] = round((b - a) / n, 4) if n > 0 else None
            out[k.replace("acc_", "delta_")] = b - a
    for k in ("acc_gfx_clk_below_host_limit_pwr", "acc_gfx_clk_below_host_limit_thm", "acc_gfx_clk_below_host_limit_total", "acc_low_utilization"):
        a, b = v0.get(k), v1.get(k)
        if isinstance(a, list) and isinstance(b, list) and a and isinstance(a[0], list):
            d = [y - x for x, y in zip(a[0], b[0]) if isinstance(x, int) and isinstance(y, int)]      # partition 0: one entry per XCD
            if d:
                out[k.replace("acc_", "delta_") + "_per_xcd"] = d
                out[k.replace("acc_", "share_") + "_mean"] = round(sum(d) / len(d) / n, 4) if n > 0 else None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--period", type=float, default=0.5)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--skip", type=float, default=3.0, help="seconds at the start of the command left out of the accumulator deltas (ramp-up, imports)")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    cmd = args.cmd[1:] if args.cmd and args.cmd[0] == "--" else args.cmd
    lim = Limiter(args.gpu)
    rows = [lim.sample(full=True)]
    stop = threading.Event()

    def loop():
        while not stop.wait(args.period):
            rows.append(lim.sample())
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    t0 = time.time()
    rc = subprocess.call(cmd) if cmd else (time.sleep(5) or 0)
    stop.set()
    th.join(timeout=5)
    rows.append(lim.sample())
    # busy window: samples with gfx activity (or power well above idle); the accumulator deltas are taken over it
    busy = [r for r in rows if r["t"] - t0 >= args.skip and isinstance(r.get("power"), dict)]

    def pw(r):
        p = r["power"]
        for k in ("socket_power", "current_socket_power", "average_socket_power"):
            if isinstance(p.get(k), (int, float)):
                return float(p[k])
        return None
    hot = [r for r in busy if (pw(r) or 0) > 600]
    win = hot if len(hot) >= 4 else busy
    summary = {"command": cmd, "rc": rc, "samples": len(rows), "window_samples": len(win)}
    if len(win) >= 2:
        summary["window_s"] = round(win[-1]["t"] - win[0]["t"], 2)
        summary["limiter"] = accumulator_shares(win[0]["violation"], win[-1]["violation"])
        ps = [pw(r) for r in win if pw(r) is not None]
        if ps:
            summary["power_w"] = {"mean": round(sum(ps) / len(ps), 1), "max": max(ps), "min": min(ps)}
        cl = []
        for r in win:
            m = r.get("metrics")
            if isinstance(m, dict):
                c = m.get("current_gfxclks") or m.get("current_gfxclk")
                if isinstance(c, list):
                    c = [v for v in c if isinstance(v, (int, float)) and 0 < v < 60000]
                    if c:
                        cl.append(sum(c) / len(c))
                elif isinstance(c, (int, float)):
                    cl.append(c)
        if cl:
            summary["gfxclk_mhz"] = {"mean": round(sum(cl) / len(cl), 1), "min": round(min(cl), 1), "max": round(max(cl), 1)}
        for k in ("temperature_hotspot", "temperature_mem", "temperature_vrsoc"):
            v = [r["metrics"][k] for r in win if isinstance(r.get("metrics"), dict) and isinstance(r["metrics"].get(k), (int, float))]
            if v:
                summary[k] = {"max": max(v), "mean": round(sum(v) / len(v), 1)}
    lim.close()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump({"summary": summary, "first_sample_full": rows[0], "rows": rows[1:]}, open(args.out, "w"), indent=1)
    print(json.dumps(summary, indent=1))
    sys.exit(rc)


if __name__ == "__main__":
    main()
