import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, tol_amd
M = tol_amd.measure_lib()
os.environ["TOLFG_PLACE_SETTLE"] = "0"
tol_amd.Batch("S10", ["tempest"], ts=4, library=M).close()
torch.zeros(1, device="cuda")
for count in (1 << 20, 32 << 20):
    keep, lost_now, lost_later = [], 0, 0
    for i in range(60 if count == 1 << 20 else 20):
        t = tol_amd.device_alloc((count,), "f64", library=M)
        t.fill_(1.0)
        torch.cuda.synchronize()
        lost_now += int((t != 1.0).sum()) > 0
        keep.append(t)
    time.sleep(0.05)
    for t in keep:
        lost_later += int((t != 1.0).sum()) > 0
    print(f"never freeing, {count} doubles: {lost_now} of {len(keep)} blocks had lost writes right after the fill, {lost_later} when looked at 50 ms after the last allocation", flush=True)
    del keep, t
    torch.cuda.synchronize()
    time.sleep(0.5)
# and: a big free, then a small block at once
big = tol_amd.device_alloc((64 << 20,), "f64", library=M)
big.fill_(3.0); torch.cuda.synchronize(); time.sleep(0.3)
lost = 0
for i in range(40):
    big = None
    big = tol_amd.device_alloc((64 << 20,), "f64", library=M)
    big.fill_(3.0); torch.cuda.synchronize(); time.sleep(0.2)      # settled by time
    big = None                                                       # free 512 MB ...
    small = tol_amd.device_alloc((1000,), "f64", library=M)          # ... and take a small block at once
    small.fill_(1.0); torch.cuda.synchronize()
    a = int((small != 1.0).sum()); time.sleep(0.02); b = int((small != 1.0).sum())
    lost += (a > 0) or (b > 0)
    del small
print(f"a small block taken right after freeing 512 MB: {lost} of 40 lost writes")
