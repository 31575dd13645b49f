#!/usr/bin/env python
"""Board power / clocks (rocm-smi, sysfs) sampled while the forward runs back to back: is the step power-limited?"""
import os, subprocess, sys, threading, time, glob
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
enc = sys.argv[1] if len(sys.argv) > 1 else "vitl"
cfg = get_config(enc)
m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
m.load_state_dict(synthetic_state_dict(cfg, seed=0)); m = m.to("cuda")
x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
stop = False
samples = []
def sample():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10)
            samples.append(r.stdout.strip()[:1500])
        except Exception as e:
            samples.append("ERR " + str(e))
        time.sleep(1.0)
print("idle:", subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True).stdout[:1500])
for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap")[:2]:
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, e)
th = threading.Thread(target=sample); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < 8:
    for _ in range(10): m.forward(x, fp32=False)
    torch.cuda.synchronize(); n += 10
dt = time.time() - t0
stop = True; th.join()
print(f"{n} forwards in {dt:.2f} s = {dt / n * 1e3:.2f} ms/clip")
for s in samples[2:6]: print(s)
