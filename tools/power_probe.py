#!/usr/bin/env python3
"""Samples the GPU's power / clock / temperature sensors (sysfs hwmon of the amdgpu device) while a command runs.
   python tools/power_probe.py OUT.json -- <command ...>      (diagnostic: is the shader clock of a bench run power-capped?)"""
import glob, json, os, subprocess, sys, threading, time

out, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]
sensors = {}
for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
    for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "temp3_input"):
        p = os.path.join(hw, name)
        if os.path.exists(p):
            sensors[hw.split("/")[4] + ":" + name] = p
samples, stop = [], False


def read(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None


def loop():
    while not stop:
        samples.append((time.time(), {k: read(p) for k, p in sensors.items()}))
        time.sleep(0.05)


t = threading.Thread(target=loop)
t.start()
t0 = time.time()
rc = subprocess.call(cmd)
stop = True
t.join()
summary = {"rc": rc, "seconds": round(time.time() - t0, 2), "n_samples": len(samples), "sensors": {}}
for k in sensors:
    v = [s[1][k] for s in samples if s[1][k] is not None]
    if v:
        summary["sensors"][k] = {"min": min(v), "max": max(v), "mean": round(sum(v) / len(v), 1), "p90": sorted(v)[int(0.9 * (len(v) - 1))]}
summary["trace_every_10th"] = [(round(s[0] - t0, 2), s[1]) for s in samples[::10]]
json.dump(summary, open(out, "w"))
print(json.dumps({k: v for k, v in summary.items() if k != "trace_every_10th"}))
sys.exit(rc)
