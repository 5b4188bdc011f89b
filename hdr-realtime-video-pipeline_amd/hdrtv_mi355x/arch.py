"""Parameter inventory of the HDRTVNet++ graph on the hot path.

Pure data: names and shapes of every tensor the three sub-networks own, in the
naming of the reference's ``state_dict`` so a checkpoint written by the reference
packs without renaming.

Reference modules restated here (structure only, no code):
  AGCM  src/models/hdrtvnet_modules/Condition_arch.py:8-35, 468-494
  LE    src/models/hdrtvnet_modules/HDRUNet3T1_arch.py:10-76, arch_util.py:60-95
  HG    src/models/hdrtvnet_modules/Hallucination_arch.py:24-36, 53-95
"""
from __future__ import annotations

NF = 32          # LE width            (hdrtvnet_torch.py:2147 nf=32)
COND_NF = 64     # LE condition width  (HDRUNet3T1_arch.py:40)
HG_NF = 64       # HG base width       (hdrtvnet_torch.py:2125 hg_nf=64)
COND_C = 6       # AGCM classifier out (hdrtvnet_torch.py:2119 cond_c=6)
GFM_NF = 64      # AGCM MLP width      (Condition_arch.py ConditionNet nf)
HG_MASK_R = 0.75       # HG_Composite_arch.py:21
HG_MASK_THRESH = 0.1   # HG_Composite_arch.py:78
BN_EPS = 1e-5
IN_EPS = 1e-5


def _conv(name, co, ci, k):
    return [(name + ".weight", (co, ci, k, k)), (name + ".bias", (co,))]


def _sft(name):
    half = NF // 2
    out = []
    for br in ("scale", "shift"):
        out += _conv(f"{name}.SFT_{br}_conv0", half, half, 1)
        out += _conv(f"{name}.SFT_{br}_conv1", NF, half, 1)
    return out


def agcm_params():
    p = []
    chans = [(0, 16, 3), (4, 32, 16), (8, 64, 32), (12, 128, 64), (16, 128, 128)]
    for i, (idx, co, ci) in enumerate(chans):
        p += _conv(f"AGCM.classifier.model.{idx}", co, ci, 1)
        if i < 4:  # InstanceNorm2d(affine) on blocks 1-4 only
            p += [(f"AGCM.classifier.model.{idx + 3}.weight", (co,)),
                  (f"AGCM.classifier.model.{idx + 3}.bias", (co,))]
    p += _conv("AGCM.classifier.model.20", COND_C, 128, 1)
    for kind in ("scale", "shift"):
        for stage, co in (("first", GFM_NF), ("HR", GFM_NF), ("last", 3)):
            p += [(f"AGCM.cond_{kind}_{stage}.weight", (co, COND_C)),
                  (f"AGCM.cond_{kind}_{stage}.bias", (co,))]
    p += _conv("AGCM.conv_first", GFM_NF, 3, 1)
    p += _conv("AGCM.HRconv", GFM_NF, GFM_NF, 1)
    p += _conv("AGCM.conv_last", 3, GFM_NF, 1)
    return p


LE_TRUNKS = (("recon_trunk1", 1), ("recon_trunk2", 1), ("recon_trunk3", 4),
             ("recon_trunk4", 1), ("recon_trunk5", 1))


def le_params():
    p = []
    p += _conv("LE.conv_first", NF, 3, 3)
    p += _sft("LE.SFT_layer1")
    p += _conv("LE.HR_conv1", NF, NF, 3)
    for i in (1, 2, 3):
        p += _conv(f"LE.down_conv{i}", NF, NF, 3)
    for trunk, n in LE_TRUNKS:
        for b in range(n):
            base = f"LE.{trunk}.{b}"
            p += _conv(base + ".conv1", NF, NF, 3)
            p += _conv(base + ".conv2", NF, NF, 3)
            p += _sft(base + ".sft1")
            p += _sft(base + ".sft2")
    for i in (1, 2, 3):
        p += _conv(f"LE.up_conv{i}.0", NF * 4, NF, 3)
    p += _sft("LE.SFT_layer2")
    p += _conv("LE.HR_conv2", NF, NF, 3)
    p += _conv("LE.conv_last", 3, NF, 3)
    p += _conv("LE.cond_first.0", COND_NF, 3, 3)
    p += _conv("LE.cond_first.2", COND_NF, COND_NF, 1)
    p += _conv("LE.cond_first.4", COND_NF, COND_NF, 1)
    half = NF // 2
    p += _conv("LE.CondNet1.0", COND_NF, COND_NF, 1)
    p += _conv("LE.CondNet1.2", COND_NF, COND_NF, 1)
    p += _conv("LE.CondNet1.4", half, COND_NF, 1)
    p += _conv("LE.CondNet2.0", COND_NF, COND_NF, 3)
    p += _conv("LE.CondNet2.2", COND_NF, COND_NF, 1)
    p += _conv("LE.CondNet2.4", half, COND_NF, 1)
    p += _conv("LE.CondNet3.0", COND_NF, COND_NF, 3)
    p += _conv("LE.CondNet3.2", COND_NF, COND_NF, 3)
    p += _conv("LE.CondNet3.4", half, COND_NF, 1)
    p += _conv("LE.CondNet4.0", COND_NF, COND_NF, 3)
    p += _conv("LE.CondNet4.2", COND_NF, COND_NF, 3)
    p += _conv("LE.CondNet4.4", half, COND_NF, 3)
    return p


def hr_params():
    """AGCM + LE: the 264 tensors of HR.pt."""
    return agcm_params() + le_params()


# HG: (name, cin, cout) of the ten conv+BN+ReLU blocks, in forward order.
HG_CONV_BLOCKS = (
    ("conv1", 3, HG_NF), ("conv2", HG_NF, 2 * HG_NF),
    ("conv3_1", 2 * HG_NF, 4 * HG_NF), ("conv3_2", 4 * HG_NF, 4 * HG_NF),
    ("conv4_1", 4 * HG_NF, 8 * HG_NF), ("conv4_2", 8 * HG_NF, 8 * HG_NF),
    ("conv5_1", 8 * HG_NF, 8 * HG_NF), ("conv5_2", 8 * HG_NF, 8 * HG_NF),
    ("conv_code1", 8 * HG_NF, 8 * HG_NF), ("conv_code2", 8 * HG_NF, 8 * HG_NF),
)
# (up name, cin, cout-before-shuffle/4, 1x1 name, 1x1 cin, 1x1 cout)
HG_UP_BLOCKS = (
    ("Up_conv1", 8 * HG_NF, 8 * HG_NF, "conv6", 16 * HG_NF, 8 * HG_NF),
    ("Up_conv2", 8 * HG_NF, 8 * HG_NF, "conv7", 16 * HG_NF, 4 * HG_NF),
    ("Up_conv3", 4 * HG_NF, 4 * HG_NF, "conv8", 8 * HG_NF, 2 * HG_NF),
    ("Up_conv4", 2 * HG_NF, 2 * HG_NF, "conv9", 4 * HG_NF, HG_NF),
    ("Up_conv5", HG_NF, HG_NF, "conv10", 2 * HG_NF, 3),
)


def hg_params(with_counters: bool = True):
    """Hallucination_Generator state_dict (keys without the ``hg.`` prefix)."""
    p = []
    for name, ci, co in HG_CONV_BLOCKS:
        p += _conv(f"{name}.0", co, ci, 3)
        p += [(f"{name}.1.weight", (co,)), (f"{name}.1.bias", (co,)),
              (f"{name}.1.running_mean", (co,)), (f"{name}.1.running_var", (co,))]
        if with_counters:
            p += [(f"{name}.1.num_batches_tracked", ())]
    for up, ci, co, one, oci, oco in HG_UP_BLOCKS:
        p += _conv(f"{up}.0", co * 4, ci, 3)
        p += _conv(one, oco, oci, 1)
    p += _conv("conv_last", 3, 6, 1)
    return p
