"""Repeat one exact (plain fp32) search N times on a fixed corpus and compare every result with the first; reports which 1M-row
blocks of the corpus changed (a wild write) and, with DUMP=1, what was written where.  The A/B harness of the round-3 compaction
race: SR_HIP_LIB=<variant .so> N_DOCS=2000000 python3 tools/micro/exact_stress_plain.py 300"""
import os, sys
import torch
import subprocess
print('host', os.uname().nodename, '|', subprocess.run('rocm-smi --showserial --showuniqueid 2>/dev/null | grep -i "serial\\|unique" | head -4', shell=True, capture_output=True, text=True).stdout.replace(chr(10), ' ; '), flush=True)
HERE = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, HERE)
from scaling_retriever_amd.scoring import DenseIndexHIP
import scaling_retriever_amd._lib as L
print("lib:", L.LIB_PATH, flush=True)
dev = torch.device("cuda", 0)
N, H, k, nq = int(os.environ.get('N_DOCS', '8841823')), 2048, 1000, 6980
g = torch.Generator(device=dev).manual_seed(1)
D = torch.empty((N, H), dtype=torch.float32, device=dev)
for r0 in range(0, N, 1 << 20):
    D[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
Q = torch.empty((nq, H), dtype=torch.float32, device=dev).normal_(0.0, 0.5 / H ** 0.5, generator=g)
Dsum0 = [D[r0:r0 + (1 << 20)].double().sum().item() for r0 in range(0, N, 1 << 20)]
Qsum0 = Q.double().sum().item()
DUMP = os.environ.get('DUMP') == '1'
D0 = D[:1 << 20].clone() if DUMP else None      # to show WHAT was written where if block 0 is corrupted
index = DenseIndexHIP(H, device=dev)
index.add_device_rows(D)
ref = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    s, i = index.search(Q, k)
    if ref is None:
        ref = (s.clone(), i.clone())
    elif not (torch.equal(s, ref[0]) and torch.equal(i, ref[1])):
        bad = (~((s == ref[0]).all(1) & (i == ref[1]).all(1))).nonzero()[:, 0]
        print("iteration", it, ":", bad.numel(), "queries differ", bad.tolist()[:10], flush=True)
Dsum1 = [D[r0:r0 + (1 << 20)].double().sum().item() for r0 in range(0, N, 1 << 20)]
print("done; 1M-row blocks of D that changed:", [b for b in range(len(Dsum0)) if Dsum0[b] != Dsum1[b]], "| Q unchanged:", Q.double().sum().item() == Qsum0, flush=True)
if DUMP:
    Di, D0i = D[:1 << 20].view(torch.int32), D0.view(torch.int32)
    rows = torch.cat([(Di[r0:r0 + 65536] != D0i[r0:r0 + 65536]).any(1).nonzero()[:, 0] + r0 for r0 in range(0, 1 << 20, 65536)])
    print("D base 0x%x; rows of block 0 with changed words: %d: %s" % (D.data_ptr(), rows.numel(), rows.tolist()[:20]), flush=True)
    shown = 0
    for r in rows.tolist()[:8]:
        cols = (Di[r] != D0i[r]).nonzero()[:, 0].tolist()
        print("  row %d: %d changed words, columns %s" % (r, len(cols), cols[:12]), flush=True)
        for c in cols[:16]:
            print("    byte offset 0x%x: 0x%08x -> 0x%08x" % ((r * H + c) * 4, D0i[r, c].item() & 0xffffffff, Di[r, c].item() & 0xffffffff), flush=True)
import ctypes
out = (ctypes.c_uint64 * 2)()
if hasattr(L.load(), "sr_debug_counters"):
    L.load().sr_debug_counters(out)
    print("compactions that met non-unique keys:", out[0], flush=True)
