#!/usr/bin/env python
"""Generate tests/golden/onet_golden.npz by running the REFERENCE's ONet modules on CPU (BASELINE config #1,
SURVEY section 8f row N4: the ONet-Opt decoder variant).

Runs only in the build container (needs /root/reference); the fixture is committed, this script is the provenance
record.  Nothing from the reference is copied: ``im2mesh.encoder.pointnet.ResnetPointnet``,
``im2mesh.onet.models.decoder.DecoderCBatchNorm`` and ``im2mesh.onet.models.OccupancyNetwork`` are imported where
they lie, loaded with seeded random weights (``pretrain/onet.pth`` is not available offline) and their inputs /
outputs saved as data.  Shims: stub modules for ``trimesh`` and the mesh-only native libs (imported eagerly by
``im2mesh/onet/__init__.py`` -> ``generation.py``) and a no-op ``Tensor.cuda``.  ``ONet/opt_defense.py`` is not
importable (argparse + cuda + torch.load at import), so its ``optimize_points`` loop (:198-238) is driven from here
with the reference's ``model.decode(p, z, c)``, ``repulsion_loss`` and ``torch.optim.Adam``.

Inputs are the clouds / draws already recorded in convonet_golden.npz (same raw clouds, same SOR + preprocess; the
encoder subset is the first 300 of the recorded 600 indices: ``pointcloud_n: 300`` in configs/onet_mn40.yaml).

    python tests/golden/make_golden_onet.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/ONet"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

for name, attrs in [("trimesh", {}), ("im2mesh.utils.libmcubes", {}),
                    ("im2mesh.utils.libsimplify", {"simplify_mesh": None}),
                    ("im2mesh.utils.libmise", {"MISE": None}),
                    ("im2mesh.utils.libkdtree", {"KDTree": None}),
                    ("im2mesh.utils.libmesh", {"check_mesh_contains": None})]:
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
torch.Tensor.cuda = lambda self, *a, **k: self

from im2mesh.encoder.pointnet import ResnetPointnet          # noqa: E402
from im2mesh.onet.models.decoder import DecoderCBatchNorm    # noqa: E402
from im2mesh.onet.models import OccupancyNetwork             # noqa: E402
from defense import repulsion_loss                            # noqa: E402

from oracle import onet_oracle as OO                          # noqa: E402

torch.set_num_threads(8)
WEIGHT_SEED = 0


def build_reference_model():
    enc = ResnetPointnet(c_dim=512, dim=3, hidden_dim=512)                 # onet_mn40.yaml:14-18
    dec = DecoderCBatchNorm(dim=3, z_dim=0, c_dim=512)                     # onet_mn40.yaml:13,18-19
    model = OccupancyNetwork(dec, enc, None, None, device=torch.device("cpu"))
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in OO.make_random_weights(WEIGHT_SEED).items()}
    print("load_state_dict:", model.load_state_dict(sd, strict=True))
    model.eval()                                                            # opt_defense.py:71
    for p in model.parameters():
        p.requires_grad = False
    return model


def ref_optimize(model, init, c, iterations, rep_weight=500.0, lr=1e-3, threshold=0.2, record=()):
    """The loop of ONet/opt_defense.py optimize_points (:198-238) around the reference model + loss."""
    x = init.clone().float()
    x.requires_grad_()
    B, K = x.shape[:2]
    z = model.get_z_from_prior((B,), sample=False)                          # :303-304 (z_dim 0: empty)
    target = torch.ones((B, K)).float() * threshold
    opt = torch.optim.Adam([x], lr=lr)
    snaps = {}
    for i in range(iterations + 1):
        occ_value = model.decode(x, z, c).logits
        occ_loss = F.binary_cross_entropy_with_logits(occ_value, target, reduction="none")
        occ_loss = torch.mean(occ_loss) * K
        rep_loss = torch.mean(repulsion_loss(x)) * rep_weight
        loss = occ_loss + rep_loss
        opt.zero_grad()
        loss.backward()
        if i in record:
            st = opt.state[x]
            snaps[i] = dict(x=x.detach().clone(), g=x.grad.detach().clone(),
                            m=st["exp_avg"].clone() if st else torch.zeros_like(x),
                            v=st["exp_avg_sq"].clone() if st else torch.zeros_like(x),
                            occ=float(occ_loss), rep=float(rep_loss))
        opt.step()
        if i in record:
            snaps[i]["x_next"] = x.detach().clone()
    return x.detach(), snaps


def ref_normalize(points):
    centroid = torch.mean(points, dim=1)
    points = points - centroid[:, None, :]
    dist = torch.sum(points ** 2, dim=2) ** 0.5
    return points / torch.max(dist, dim=1)[0][:, None, None]


def main():
    g = np.load(os.path.join(HERE, "convonet_golden.npz"))
    model = build_reference_model()
    ow = OO.to_torch(OO.make_random_weights(WEIGHT_SEED))
    out = {}
    proc = [g["proc_pad"][b, :g["proc_len"][b]] for b in range(4)]
    sel = torch.from_numpy(np.stack([proc[b][g["sel_idx"][b][:300]] for b in range(4)]))     # [4,300,3]
    out["sel"] = sel.numpy()
    init = torch.from_numpy(g["init_points"])                                                # [4,1024,3]

    # encoder
    c = model.encode_inputs(sel)
    out["c"] = c.numpy()
    c_o, stages = OO.encode_latent(ow, sel, return_stages=True)
    print("encoder oracle vs reference: max |dc| %.3e (|c| max %.3f)" % (float((c_o - c).abs().max()), float(c.abs().max())))
    out["enc_stage_max0"] = torch.stack([s[0].max(dim=0).values for s in stages]).numpy()    # [5,512] cloud 0

    # decoder logits + d(sum logits)/dp
    p = init[:2].clone().requires_grad_()
    z = model.get_z_from_prior((2,), sample=False)
    logits = model.decode(p, z, c[:2]).logits
    logits.sum().backward()
    out["dec_logits"] = logits.detach().numpy()
    out["dec_dlogit_dp"] = p.grad.numpy()
    lo = OO.decode_logits(ow, init[:2], c[:2])
    print("decoder oracle vs reference: max |dlogit| %.3e (|logit| max %.3f)" % (float((lo - logits).abs().max()), float(logits.abs().max())))

    # trajectories with Adam state (B = 2)
    x10, snaps = ref_optimize(model, init[:2], c[:2], iterations=10, record=(0, 1, 9))
    for i, s in snaps.items():
        for k in ("x", "g", "m", "v", "x_next"):
            out[f"traj{i}_{k}"] = s[k].numpy()
        out[f"traj{i}_loss"] = np.array([s["occ"], s["rep"]], np.float64)
    xo = OO.optimize_points(ow, init[:2], c[:2], iterations=9, normalize=False)
    d = (xo - snaps[9]["x_next"]).norm(dim=-1)
    print("oracle vs reference after 10 steps: max %.3e" % float(d.max()))

    # end to end, 4 clouds, 11 steps, normalised
    x11, _ = ref_optimize(model, init, c, iterations=10)
    out["e2e10_out"] = ref_normalize(x11).numpy()
    path = os.path.join(HERE, "onet_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
