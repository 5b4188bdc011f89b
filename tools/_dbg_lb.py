import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
import numpy as np, torch
from oracle import letterbox_oracle as L
from hdrtv_mi355x.processor import HDRTVNetMI355X
p = HDRTVNetMI355X(os.path.join(REPO, "tests/golden/hr_weights.hdrw"), use_hg=False, warmup_passes=0)
for src_hw, dst_hw in [((100, 180), (64, 96)), ((72, 128), (180, 320))]:
    rng = np.random.default_rng(src_hw[0] * 1000 + dst_hw[1])
    f = rng.integers(0, 256, (*src_hw, 3), dtype=np.uint8)
    want = L.letterbox_bgr(f, dst_hw[1], dst_hw[0])
    src = torch.from_numpy(f).cuda(); dst = torch.full((*dst_hw, 3), 7, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p._chk(p._lib.hdrtv_letterbox_u8(p._ctx, st, src.data_ptr(), src_hw[0], src_hw[1], dst.data_ptr(), dst_hw[0], dst_hw[1]), "lb")
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    d = got.astype(int) - want.astype(int)
    nz = np.argwhere(d != 0)
    print(src_hw, dst_hw, L.geometry(src_hw[1], src_hw[0], dst_hw[1], dst_hw[0]), "mismatch", len(nz), "max", np.abs(d).max())
    print(nz[:8].tolist(), [int(d[tuple(i)]) for i in nz[:8]])
