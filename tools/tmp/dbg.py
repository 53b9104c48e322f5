import sys, atexit, faulthandler
faulthandler.enable()
sys.path.insert(0, '.')
atexit.register(lambda: print("atexit reached", file=sys.stderr, flush=True))
import bench, torch
torch.cuda.set_device(0)
print("start", file=sys.stderr, flush=True)
job = bench.SystemJob(0, 7, host_copy=False, width=1280, height=720, cell=15)
print("job built", file=sys.stderr, flush=True)
for i in range(900):
    job.step()
    if i % 50 == 0:
        print(i, list(job.ar.state()), file=sys.stderr, flush=True)
print("done", file=sys.stderr, flush=True)
