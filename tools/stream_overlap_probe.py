"""development aid: do kernels of two HIP streams of one process run side by side on this device?  A long single-block
spin kernel on stream A, a short kernel on stream B launched right behind it: if B ends long before A does, they overlapped."""
import torch
dev = torch.device("cuda")
b = torch.zeros(1 << 16, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for trial in range(3):
    torch.cuda.synchronize()
    ea0, ea1, eb0, eb1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.cuda.stream(sa):
        ea0.record(); torch.cuda._sleep(2000000); ea1.record()
    with torch.cuda.stream(sb):
        eb0.record(); b.add_(1.0); eb1.record()
    torch.cuda.synchronize()
    print("A: %.1f us; B starts %.1f us after A's start, ends %.1f us after A's start" % (ea0.elapsed_time(ea1) * 1e3, ea0.elapsed_time(eb0) * 1e3, ea0.elapsed_time(eb1) * 1e3))
