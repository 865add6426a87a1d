#!/usr/bin/env python
"""Package power of every card (hwmon power1_average, watts) at 4 Hz for argv[1] seconds; never touches the GPU runtime.
Run beside a GPU program:  python tools/probe/power_sample.py 10 > power.txt &"""
import glob, sys, time
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) or sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
t_end = time.time() + float(sys.argv[1])
t0 = time.time()
while time.time() < t_end:
    pw = []
    for h in hw:
        try:
            pw.append(int(open(h).read()) / 1e6)
        except Exception:
            pw.append(0.0)
    print(f"t={time.time() - t0:5.2f} s  max {max(pw):6.0f} W   all: " + " ".join(f"{v:.0f}" for v in pw), flush=True)
    time.sleep(0.25)
