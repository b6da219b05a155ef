import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sameold_amd as sa
C, T = 4096, 220500
x = sa.synth_afsk(C, T, 22050, seed=1); torch.cuda.synchronize()
rx = sa.SameReceiverBuilder(22050).build_batch(C, link_only=bool(int(os.environ.get("LINK", "0"))))
rx.set_kernel_timing(bool(int(os.environ.get("KT", "1"))))
rx.process_tensor(x); rx.sync(); rx.poll_events_np()
t0 = time.perf_counter()
for i in range(4):
    rx.process_tensor(x)
    print("pass", i, "returned at %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    if os.environ.get("SLEEP"):
        time.sleep(float(os.environ["SLEEP"]))
rx.sync()
print("done at %.1f ms" % ((time.perf_counter() - t0) * 1e3))
