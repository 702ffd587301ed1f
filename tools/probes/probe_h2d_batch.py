import sys, time, torch
sys.path.insert(0, "soft-intro-vae-pytorch_amd")
from sivae_hip import ops
x = torch.randint(0, 255, (128, 256, 256, 3), dtype=torch.uint8).pin_memory()
flip = torch.zeros(128, dtype=torch.int32, device="cuda")
for _ in range(3):
    d = x.to("cuda", non_blocking=True); y = ops.u8_to_f32(d, flip, nhwc=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    d = x.to("cuda", non_blocking=True); y = ops.u8_to_f32(d, flip, nhwc=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("H2D of 128 uint8 256x256x3 images (25.2 MB, pinned) + u8 -> f32 NCHW on the device: %.3f ms per batch = %.1f GB/s host->device" % (dt * 1e3, x.numel() / dt / 1e9))
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): y = ops.u8_to_f32(d, flip, nhwc=True)
e.record(); torch.cuda.synchronize()
print("u8 -> f32 kernel alone: %.3f ms" % (s.elapsed_time(e) / 20))
