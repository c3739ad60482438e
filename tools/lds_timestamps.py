import ctypes as C, os, sys, collections
os.environ["SBGM_LDS_TS"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from sbgm_danra_amd import _native as N
L = N.lib()
B, Cin, Cout, H = 32, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tile = (int(sys.argv[4]), int(sys.argv[5])); db = int(sys.argv[6])
x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
packed = torch.empty(L.sbgm_conv_wino_packed_numel(Cout, Cin), device="cuda")
N.check(L.sbgm_conv_wino_pack_weight(w.data_ptr(), packed.data_ptr(), Cout, Cin, Cin, N.stream()))
out = torch.empty(B, H, H, Cout, device="cuda")
a = N.ConvArgs(x.data_ptr(), packed.data_ptr(), out.data_ptr(), None, None, None, None, B, H, H, Cin, Cout, 3, 3, 1, 1, 0, 0, tile[0], tile[1], 0, 0, 3 | db, 0, 0, 0, None, 0)
for _ in range(3): N.check(L.sbgm_conv2d_fwd(C.byref(a), N.stream()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record(); N.check(L.sbgm_conv2d_fwd(C.byref(a), N.stream())); e1.record(); torch.cuda.synchronize()
print("kernel ms", e0.elapsed_time(e1))
ptr, tiles = open("/tmp/sbgm_lds_ts_ptr").read().split(); ptr, tiles = int(ptr), int(tiles)
hip = C.CDLL("libamdhip64.so")
buf = np.zeros(tiles * 6, dtype=np.uint64)
rc = hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(buf.nbytes), 2)
assert rc == 0, rc
t = buf.reshape(tiles, 6)
t0, t1, t2, t3 = (t[:, i].astype(np.int64) for i in range(4))
hw, xcc = t[:, 4], t[:, 5]
cu = ((hw >> 8) & 0xF).astype(np.int64); se = ((hw >> 13) & 0x7).astype(np.int64); sh = ((hw >> 12) & 1).astype(np.int64)
key = (xcc.astype(np.int64) * 8 + se) * 32 + sh * 16 + cu
print("tiles", tiles, "distinct CUs", len(set(key.tolist())))
print("per-WG cycles: prologue %.0f  loop %.0f  epilogue %.0f  total %.0f" % ((t1 - t0).mean(), (t2 - t1).mean(), (t3 - t2).mean(), (t3 - t0).mean()))
span = t3.max() - t0.min()
print("kernel span cycles", span)
# per CU: busy union, number of WGs, gaps
per = collections.defaultdict(list)
for k, a_, b_ in zip(key.tolist(), t0.tolist(), t3.tolist()): per[k].append((a_, b_))
conc, gaps = [], []
for k, iv in per.items():
    iv.sort()
    tot = sum(b_ - a_ for a_, b_ in iv)
    conc.append(tot / (iv[-1][1] - iv[0][0]))
print("WGs per CU avg %.1f, average concurrency (sum of lifetimes / CU active span) %.2f" % (tiles / len(per), float(np.mean(conc))))
first_start = np.array([min(a_ for a_, _ in iv) for iv in per.values()]); last_end = np.array([max(b_ for _, b_ in iv) for iv in per.values()])
print("CU start skew %.0f cycles, end skew %.0f cycles" % (first_start.max() - first_start.min(), last_end.max() - last_end.min()))
