"""CPU-only: replicate preprocess_bwd's conic -> cov2D -> Sigma -> (scales, rotations) chain in torch fp32 and compare each
stage with autograd through the oracle graph (fp32 and fp64) on needle-shaped splats."""
import sys, torch
sys.path[:0] = ['.', 'bundle-adjusting-gaussian-splatting_amd', 'tests']      # run from the repo root: python tests/analysis_needle_chain.py
from parity import run_oracle
from oracle import raster_oracle as O
from scenes import make_case, rel_err, oracle_settings
P = 600
for mult in (1.0, 10.0, 30.0):
    scene, cam = make_case(P, 144, 112, 2.0, 0, seed=41)
    scene["scales"] = scene["scales"] * torch.tensor([mult, 0.02, 0.02])
    g = torch.randn(3, 112, 144, generator=torch.Generator().manual_seed(1))
    st32, gr32 = run_oracle(scene, cam, 0, g, torch.float32)
    res = {}
    for dt in (torch.float32, torch.float64):
        s = oracle_settings(cam, 0)
        leaf = {k: v.to(dt).clone().requires_grad_(True) for k, v in scene.items()}
        pre = O.preprocess(leaf["means3D"], torch.zeros(P, 3, dtype=dt), torch.zeros(3, dtype=dt), leaf["shs"], None, leaf["opacities"],
                           leaf["scales"], leaf["rotations"], None, s, dtype=dt, discrete=O.discrete_of(st32))
        G = pre.extras["_graph"]
        gcon = gr32["_2d"]["conic"].to(dt)
        outs = list(G["cov"]) + list(G["sigma"]) + list(G["A"]) + [leaf["scales"], leaf["rotations"]]
        gr = torch.autograd.grad([pre.conic], outs, [gcon], allow_unused=True)
        res[dt] = dict(dcov=torch.stack(gr[0:3], 1), dsig=torch.stack(gr[3:9], 1), dA=torch.stack(gr[9:15], 1), dscale=gr[15], drot=gr[16],
                       G=G, leaf=leaf)
    r32, r64 = res[torch.float32], res[torch.float64]
    idx = r32["G"]["idx"]
    # ---- kernel formulas, fp32
    cxx, cxy, cyy = [t.detach() for t in r32["G"]["cov"]]
    a00, a01, a02, a10, a11, a12 = [t.detach() for t in r32["G"]["A"]]
    gA, gB, gC = [gr32["_2d"]["conic"][idx, i] for i in range(3)]
    det = cxx * cyy - cxy * cxy
    di = 1.0 / det; di2 = di * di
    if "--structured" in sys.argv:
        T = gA * cyy - gB * cxy + gC * cxx
        gdet = -(T * di2)
        dcxx = gC * di + gdet * cyy
        dcyy = gA * di + gdet * cxx
        dcxy = -(gB * di) - 2. * (gdet * cxy)
    else:
        dcxx = di2 * (-gA * cyy * cyy + gB * cxy * cyy - gC * cxy * cxy)
        dcyy = di2 * (-gA * cxy * cxy + gB * cxy * cxx - gC * cxx * cxx)
        dcxy = di2 * (2. * gA * cyy * cxy + 2. * gC * cxx * cxy) - gB * (di + 2. * cxy * cxy * di2)
    k_dcov = torch.stack([dcxx, dcxy, dcyy], 1)
    print("mult", mult)
    print("  dcov : kernel-form vs ag64 %.2e | ag32 vs ag64 %.2e" % (rel_err(k_dcov, r64["dcov"]), rel_err(r32["dcov"], r64["dcov"])))
    def sig_from(dcxx, dcxy, dcyy):
        g0 = dcxx * a00 * a00 + dcxy * a00 * a10 + dcyy * a10 * a10
        g3 = dcxx * a01 * a01 + dcxy * a01 * a11 + dcyy * a11 * a11
        g5 = dcxx * a02 * a02 + dcxy * a02 * a12 + dcyy * a12 * a12
        g1 = 2. * dcxx * a00 * a01 + dcxy * (a00 * a11 + a01 * a10) + 2. * dcyy * a10 * a11
        g2 = 2. * dcxx * a00 * a02 + dcxy * (a00 * a12 + a02 * a10) + 2. * dcyy * a10 * a12
        g4 = 2. * dcxx * a01 * a02 + dcxy * (a01 * a12 + a02 * a11) + 2. * dcyy * a11 * a12
        return torch.stack([g0, g1, g2, g3, g4, g5], 1)
    k_sig = sig_from(dcxx, dcxy, dcyy)
    k_sig_ag = sig_from(*[r32["dcov"][:, i] for i in range(3)])
    print("  dSig : kernel vs ag64 %.2e | kernel-from-ag32-dcov vs ag64 %.2e | ag32 vs ag64 %.2e" %
          (rel_err(k_sig, r64["dsig"]), rel_err(k_sig_ag, r64["dsig"]), rel_err(r32["dsig"], r64["dsig"])))
    print("  final: ag32 scales %.2e rot %.2e (vs ag64)" % (rel_err(r32["dscale"], r64["dscale"]), rel_err(r32["drot"], r64["drot"])))
    b = {}
    c0, c1, c2, c3, c4, c5 = [t.detach() for t in r32["G"]["sigma"]]
    b00 = a00 * c0 + a01 * c1 + a02 * c2; b01 = a00 * c1 + a01 * c3 + a02 * c4; b02 = a00 * c2 + a01 * c4 + a02 * c5
    b10 = a10 * c0 + a11 * c1 + a12 * c2; b11 = a10 * c1 + a11 * c3 + a12 * c4; b12 = a10 * c2 + a11 * c4 + a12 * c5
    dA = torch.stack([2. * dcxx * b00 + dcxy * b10, 2. * dcxx * b01 + dcxy * b11, 2. * dcxx * b02 + dcxy * b12,
                      2. * dcyy * b10 + dcxy * b00, 2. * dcyy * b11 + dcxy * b01, 2. * dcyy * b12 + dcxy * b02], 1)
    print("  dA   : kernel vs ag64 %.2e | ag32 vs ag64 %.2e" % (rel_err(dA, r64["dA"]), rel_err(r32["dA"], r64["dA"])))
    # Sigma -> scales / rotations
    sc = scene["scales"][idx] * 1.0; q = scene["rotations"][idx]
    s0, s1, s2 = sc[:, 0], sc[:, 1], sc[:, 2]; qr, qx, qy, qz = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    r00 = 1.0 - 2.0 * (qy * qy + qz * qz); r01 = 2.0 * (qx * qy - qr * qz); r02 = 2.0 * (qx * qz + qr * qy)
    r10 = 2.0 * (qx * qy + qr * qz); r11 = 1.0 - 2.0 * (qx * qx + qz * qz); r12 = 2.0 * (qy * qz - qr * qx)
    r20 = 2.0 * (qx * qz - qr * qy); r21 = 2.0 * (qy * qz + qr * qx); r22 = 1.0 - 2.0 * (qx * qx + qy * qy)
    gc = k_sig
    S00, S01, S02, S11, S12, S22 = 2. * gc[:, 0], gc[:, 1], gc[:, 2], 2. * gc[:, 3], gc[:, 4], 2. * gc[:, 5]
    l00, l01, l02 = r00 * s0, r01 * s1, r02 * s2
    l10, l11, l12 = r10 * s0, r11 * s1, r12 * s2
    l20, l21, l22 = r20 * s0, r21 * s1, r22 * s2
    dl00 = S00 * l00 + S01 * l10 + S02 * l20; dl01 = S00 * l01 + S01 * l11 + S02 * l21; dl02 = S00 * l02 + S01 * l12 + S02 * l22
    dl10 = S01 * l00 + S11 * l10 + S12 * l20; dl11 = S01 * l01 + S11 * l11 + S12 * l21; dl12 = S01 * l02 + S11 * l12 + S12 * l22
    dl20 = S02 * l00 + S12 * l10 + S22 * l20; dl21 = S02 * l01 + S12 * l11 + S22 * l21; dl22 = S02 * l02 + S12 * l12 + S22 * l22
    gs = torch.stack([dl00 * r00 + dl10 * r10 + dl20 * r20, dl01 * r01 + dl11 * r11 + dl21 * r21, dl02 * r02 + dl12 * r12 + dl22 * r22], 1)
    d00, d01, d02 = dl00 * s0, dl01 * s1, dl02 * s2
    d10, d11, d12 = dl10 * s0, dl11 * s1, dl12 * s2
    d20, d21, d22 = dl20 * s0, dl21 * s1, dl22 * s2
    gq = torch.stack([2. * (-qz * d01 + qy * d02 + qz * d10 - qx * d12 - qy * d20 + qx * d21),
        2. * (qy * d01 + qz * d02 + qy * d10 - 2. * qx * d11 - qr * d12 + qz * d20 + qr * d21 - 2. * qx * d22),
        2. * (-2. * qy * d00 + qx * d01 + qr * d02 + qx * d10 + qz * d12 - qr * d20 + qz * d21 - 2. * qy * d22),
        2. * (-2. * qz * d00 - qr * d01 + qx * d02 + qr * d10 - 2. * qz * d11 + qy * d12 + qx * d20 + qy * d21)], 1)
    print("  scale: kernel vs ag64 %.2e | rot: kernel vs ag64 %.2e" % (rel_err(gs, r64["dscale"][idx]), rel_err(gq, r64["drot"][idx])))
