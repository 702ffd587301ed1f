"""Shader clock under a sustained F(4x4,3x3) load (GPU box): samples the amdgpu sysfs clock files while a conv loops."""
import glob, os, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
import torch
from sivae_hip import ops

def read_clocks():
    out = {}
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk") + glob.glob("/sys/class/drm/card*/device/pp_dpm_mclk"):
        try:
            cur = [l.strip() for l in open(f).read().splitlines() if l.strip().endswith("*")]
            out[f.split("/")[4] + "/" + os.path.basename(f)] = cur
        except OSError as e:
            out[f] = str(e)
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq*_input") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        try:
            out[f.split("/")[4] + "/" + os.path.basename(f)] = open(f).read().strip()
        except OSError as e:
            out[f] = str(e)
    return out

print("idle", read_clocks())
for (Ci, Co, H) in [(512, 512, 32), (64, 64, 256)]:
    B = 128
    x = torch.randn(B, Ci, H, H, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
    wq = ops.PackedW(w, 0)
    stop = [False]
    samples = []
    def sampler():
        while not stop[0]:
            samples.append(read_clocks()); time.sleep(0.25)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 2.5:
        for _ in range(50):
            ops.conv2d_fwd(x, wq, Co, 3, want_stats=True)
        torch.cuda.synchronize(); n += 50
    dt = time.time() - t0
    stop[0] = True; th.join()
    fl = 2.0 * B * H * H * Ci * Co * 9 / 4
    print("%d->%d@%d: %.3f ms per launch, %.1f TF/s executed" % (Ci, Co, H, dt / n * 1e3, fl * n / dt / 1e12))
    for s_ in samples[2:8]:
        print("  ", s_)
