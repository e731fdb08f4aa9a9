"""Device time per minibatch step with device-resident indices."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate
dev = torch.device("cuda", 0)
class _M: K, device = 2, dev
data = simulate(_M, 400, 1000, 1, 14, seed=1000, params=TEST_PARAMS)
if os.environ.get("OFFSETS") == "hist":  # the 50-value histogram of bench.py --offsets hist
    from tapqir_amd.utils.dataset import CosmosDataset
    s_ = torch.arange(70.0, 120.0)
    w_ = torch.minimum(s_ - 69.0, 120.0 - s_)
    data = CosmosDataset(data.images, data.xy, data.is_ontarget, offset_samples=s_, offset_weights=w_ / w_.sum())
eng = CosmosEngine(data, K=2, device=dev, seed=7)
eng.layout.set_constrained(eng.params, initial_values(eng, data))
if os.environ.get("TRAINED"):  # the parameter regime of a converged fit (scripts/site_trained.py)
    for _ in range(int(os.environ["TRAINED"])):
        eng.step()
    eng.join()
g = torch.Generator().manual_seed(0)
nb, fb = int(os.environ.get("NB", 10)), int(os.environ.get("FB", 512))
didx = [(torch.randperm(400, generator=g)[:nb].to(dev, torch.int32), torch.randperm(1000, generator=g)[:fb].to(dev, torch.int32)) for _ in range(200)]
for nd, fd in didx[:20]:
    eng.step(nd, fd)
torch.cuda.synchronize()
if os.environ.get("SUBSAMPLED"):  # subsamples drawn on the device by the tail workgroup of the previous launch
    gen = torch.Generator().manual_seed(0)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(len(didx)):
            assert eng.step_subsampled(nb, fb, gen)
        torch.cuda.synchronize()
        print(f"{nb}x{fb}, device-drawn subsamples: {1e6 * (time.perf_counter() - t0) / len(didx):.1f} us/step", flush=True)
for rep in range(3):
    if os.environ.get("STAMPS"):
        torch.cuda.synchronize()
        eng._sync[4 + 32:4 + 48].zero_()  # maxima over the workgroups (stamps build)
    t0 = time.perf_counter()
    for nd, fd in didx:
        eng.step(nd, fd)
    torch.cuda.synchronize()
    print(f"{nb}x{fb}: {1e6 * (time.perf_counter() - t0) / len(didx):.1f} us/step", flush=True)
if os.environ.get("STAMPS"):
    torch.cuda.synchronize()
    st = eng._sync[4:32].cpu().view(torch.int64)
    print("ticket barrier %.1f us, to phase-1 start %.1f us" % ((int(st[6]) - int(st[0])) / 100.0, (int(st[7]) - int(st[0])) / 100.0))
    d = [(int(st[i + 1]) - int(st[i])) / 100.0 for i in range(5)]
    print("stamps (us): catchup %.1f  sites %.1f  wait %.1f  pixel %.1f  unit %.1f  total %.1f" % (*d, (int(st[5]) - int(st[0])) / 100.0))
    print("  inside the tail: per-AOI sums done %.1f, cross-unit sums (gsum) done %.1f" % ((int(st[13]) - int(st[0])) / 100.0, (int(st[12]) - int(st[0])) / 100.0))
    def where(v):
        return "xcc %d cu %02x block %d ticket %d" % ((v >> 28) & 15, (v >> 20) & 255, (v >> 10) & 1023, v & 1023)
    w = eng._sync[4 + 44:4 + 48].cpu().view(torch.int64)
    print("slowest workgroup: %.1f us at %s;  tail workgroup at %s" % ((int(w[0]) >> 32) / 100.0, where(int(w[0]) & 0xffffffff), where(int(w[1]))))
    mx = eng._sync[4 + 32:4 + 44].cpu().view(torch.int64)
    print("maxima over the workgroups (us): catchup %.1f  sites %.1f  wait %.1f  pixel %.1f  unit %.1f  total %.1f" % tuple(int(v) / 100.0 for v in mx))
    if os.environ.get("CATCHUP"):  # inside the catch-up phase of workgroup STAMPS, thread 0
        sw = eng._sync[4 + 48:4 + 56].cpu().view(torch.int64)
        print("catch-up (us since its start): table built %.1f, barrier %.1f, first replay pass %.1f, second %.1f" %
              tuple((int(v) - int(st[7])) / 100.0 for v in sw))
    if os.environ.get("SITES"):  # -DTQ_MB_STAMPS_SITES=1: gradient of global site s done; =2: its draw done (us since the tail's start)
        sw = eng._sync[4 + 48:4 + 56].cpu().view(torch.int64)
        print("global sites (%s) done at" % ("gradient" if os.environ["SITES"] == "1" else "draw"),
              " ".join("%.1f" % ((int(v) - int(st[8])) / 100.0) for v in sw), "us after the tail's start")
    t = [(int(st[i]) - int(st[0])) / 100.0 for i in (8, 9, 10, 11)]
    print("tail workgroup (us since stamp 0 of block %s): start %.1f  sums+global grads done %.1f  adam done %.1f  flag %.1f" % (os.environ.get("STAMPS"), *t))
