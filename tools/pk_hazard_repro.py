#!/usr/bin/env python3
"""DESIGN.md lesson 12, second stage: what exactly makes packed-fp32 accumulators go wrong next to a bf16-MFMA kernel?

  python tools/pk_hazard_repro.py [launches=200] [parts=1,1b,2]

Part 1 — the MINIMAL victim (tools/pk_hazard_repro.hip: no LDS, no atomics, per-thread accumulators written straight to
          memory; compared BIT FOR BIT with its own result when it runs alone) beside library GEMMs (torch.matmul: code that
          is not ours) and beside the synthetic register-only MFMA loops, with float data and with small-integer data.
Part 2 — the REAL victim (glowtts_actnorm_invconv_bwd from the library named by GLOWTTS_HIP_LIB; build the failing variant
          with tools/exp_build.sh flows.hip "slp:-fslp-vectorize") beside a bf16 library GEMM, (a) both on the whole chip,
          (b) on DISJOINT halves of the compute units (CU-masked streams): same power draw and memory pressure, no shared SIMD.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "glow-tts-train_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def masked_stream(lib, lo, hi, n_cu=256):
    words = (ctypes.c_uint * (n_cu // 32))()
    for cu in range(lo, hi):
        words[cu // 32] |= 1 << (cu % 32)
    ptr = lib.pk_masked_stream(words, n_cu // 32)
    assert ptr, "hipExtStreamCreateWithCUMask failed"
    return torch.cuda.ExternalStream(ptr)


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    parts = (sys.argv[2] if len(sys.argv) > 2 else "1,1b,2").split(",")
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libpk_hazard_repro.bin"))
    lib.pk_masked_stream.restype = ctypes.c_void_p
    lib.pk_victim.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    lib.pk_aggressor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
    dev = "cuda"
    torch.zeros(1, device=dev)
    a16 = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
    a32 = torch.randn(1024, 1024, device=dev)
    sink = torch.zeros(4 * (64 + (1 << 20)) + 1024, device=dev)       # kind 7 streams 16 MB of it
    side = torch.cuda.Stream()
    half_a, half_b = masked_stream(lib, 0, 128), masked_stream(lib, 128, 256)

    def gemm16():
        torch.matmul(a16, a16)

    def gemm32():
        torch.matmul(a32, a32)

    def synth(kind):
        return lambda: [lib.pk_aggressor(kind, 288, sink.data_ptr(), 160, 7, torch.cuda.current_stream().cuda_stream) for _ in range(3)]

    aggressors = [("nothing", None), ("torch bf16 GEMM 2048^3", gemm16), ("torch fp32 GEMM 1024^3", gemm32),
                  ("synthetic bf16 MFMA loop (registers only)", synth(1)), ("synthetic bf16 MFMA loop + LDS reads", synth(4))]
    more = [("synthetic bf16 MFMA loop, 256 VGPRs allocated", synth(5)), ("synthetic bf16 MFMA loop, 512 registers", synth(6)),
            ("synthetic bf16 MFMA loop + streaming loads", synth(7)), ("synthetic fp32 VALU loop, 256 VGPRs, no MFMA", synth(8)),
            ("synthetic bf16 MFMA loop + LDS reads+writes", synth(9)),
            ("synthetic LDS reads + ds_write_b64, fp32 VALU, no MFMA", synth(10)), ("synthetic bf16 MFMA loop + ds_write_b32", synth(11)),
            ("synthetic bf16 MFMA loop + ds_write_b128", synth(12)), ("synthetic ds_write_b64 only", synth(13)),
            ("synthetic bf16 MFMA loop + ds_write2_b32", synth(14)), ("synthetic bf16 MFMA loop + ds_write_b16", synth(15)),
            ("synthetic bf16 MFMA loop + ds_write_b96", synth(16)), ("synthetic FP32 MFMA loop + ds_write_b64", synth(17)),
            ("synthetic bf16 MFMA loop + ds_write2_b32 adjacent", synth(18)), ("synthetic bf16 MFMA loop + ds_write2st64_b64", synth(19))]

    print("== part 1: minimal victim (bitwise against its own solo run)")
    groups = 40
    for nb in ((128, 248) if "1" in parts else ()):
        n_items = nb * groups
        for data in ("float", "small-int"):
            g = torch.Generator(device=dev).manual_seed(nb)
            if data == "float":
                x = torch.randn(n_items * 16, device=dev, generator=g)
                gz = torch.randn(n_items * 16, device=dev, generator=g)
                m = torch.ones(n_items * 4, device=dev)
            else:
                x = torch.randint(-2, 3, (n_items * 16,), device=dev, generator=g).float()
                gz = torch.randint(-3, 4, (n_items * 16,), device=dev, generator=g).float()
                m = torch.randint(0, 2, (n_items * 4,), device=dev, generator=g).float()
            for pk in (1, 0):
                out = torch.empty(groups * 256 * 24, device=dev)

                def victim():
                    out.fill_(float("nan"))
                    lib.pk_victim(pk, x.data_ptr(), gz.data_ptr(), m.data_ptr(), out.data_ptr(), groups, n_items, nb,
                                  torch.cuda.current_stream().cuda_stream)

                victim()
                torch.cuda.synchronize()
                ref = out.clone()
                for name, fn in aggressors:
                    bad, lanes = 0, torch.zeros(64, dtype=torch.long, device=dev)
                    for _ in range(launches):
                        torch.cuda.synchronize()
                        if fn is not None:
                            with torch.cuda.stream(side):
                                fn()
                        victim()
                        torch.cuda.synchronize()
                        ne = (out.view(torch.int32) != ref.view(torch.int32)).view(groups, 4, 64, 24)
                        if bool(ne.any()):
                            bad += 1
                            lanes += ne.any(3).sum((0, 1))
                    print(f"items/group {nb:3d} {data:9s} victim {'packed' if pk else 'scalar'} beside {name:42s}: "
                          f"{bad:4d} / {launches} launches differ" + (f"  lanes {lanes.tolist()}" if bad else ""), flush=True)

    print("== part 1b: the loop of actnorm_invconv_bwd itself, copied into the stand-alone library (no LDS, no atomics): built with"
          " and without the SLP vectorizer, uniform multipliers as SGPR pairs or forced into VGPRs; bitwise against its solo run")
    for build, libname in ((("SLP build (v_pk_* with SGPR-pair operands)", "libpk_hazard_repro_slp.bin"),
                            ("-fno-slp-vectorize build", "libpk_hazard_repro.bin")) if "1b" in parts else
                           (("SLP build (v_pk_* with SGPR-pair operands)", "libpk_hazard_repro_slp.bin"),) if "1b-slp" in parts else ()):
        vl = ctypes.CDLL(os.path.join(ROOT, "tools", libname))
        vl.pk_victim_real.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 8 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
        nb = 248
        n_items = nb * groups
        g = torch.Generator(device=dev).manual_seed(5)
        x = torch.randn(n_items * 16, device=dev, generator=g)
        gz = torch.randn(n_items * 16, device=dev, generator=g)
        m = torch.ones(n_items * 4, device=dev)
        wq = torch.linalg.qr(torch.randn(4, 4))[0].to(dev).contiguous()
        logs, bias = torch.randn(groups * 4, device=dev) * 0.1, torch.randn(groups * 4, device=dev) * 0.1
        dxo = torch.empty(n_items * 16, device=dev)
        out = torch.empty(groups * 256 * 24, device=dev)
        for in_vgpr in ((0,) if "1b-slp" in parts else (0, 1)):
            def victim(stream=None):
                out.fill_(float("nan"))
                vl.pk_victim_real(in_vgpr, x.data_ptr(), gz.data_ptr(), m.data_ptr(), wq.data_ptr(), logs.data_ptr(), bias.data_ptr(),
                                  dxo.data_ptr(), out.data_ptr(), groups, n_items, nb, torch.cuda.current_stream().cuda_stream)

            victim()
            torch.cuda.synchronize()
            ref, ref_dx = out.clone(), dxo.clone()
            cases = [(name, fn, None, side) for name, fn in aggressors + more]
            cases.append(("torch bf16 GEMM, DISJOINT CU halves", gemm16, half_b, half_a))
            for name, fn, vstream, astream in cases:
                bad, lanes, accs = 0, torch.zeros(64, dtype=torch.long, device=dev), torch.zeros(24, dtype=torch.long, device=dev)
                for _ in range(launches):
                    torch.cuda.synchronize()
                    if fn is not None:
                        with torch.cuda.stream(astream):
                            fn()
                    if vstream is None:
                        victim()
                    else:
                        with torch.cuda.stream(vstream):
                            victim()
                    torch.cuda.synchronize()
                    ne = (out.view(torch.int32) != ref.view(torch.int32)).view(groups, 4, 64, 24)
                    if bool(ne.any()) or not torch.equal(dxo, ref_dx):
                        bad += 1
                        lanes += ne.any(3).sum((0, 1))
                        accs += ne.sum((0, 1, 2))
                print(f"{build:45s} multipliers in {'VGPRs' if in_vgpr else 'SGPRs'} beside {name:42s}: {bad:4d} / {launches} differ"
                      + (f"  lanes {lanes.tolist()} accumulators {accs.tolist()}" if bad else ""), flush=True)

    if "2" not in parts:
        return
    print("== part 2: the real victim, library", os.environ.get("GLOWTTS_HIP_LIB", "(default)"))
    from glow_tts_train import _hip
    from glow_tts_train._hip import call, ptr

    _hip.load()
    B, C, T = 8, 160, 124
    torch.manual_seed(0)
    x, dz = torch.randn(B, C, T, device=dev), torch.randn(B, C, T, device=dev)
    m = torch.ones(B, T, device=dev)
    logs, bias = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
    w = torch.linalg.qr(torch.randn(4, 4))[0].to(dev).contiguous()
    winv = torch.linalg.inv(w).contiguous()
    xlen, dld = m.sum(1), torch.randn(B, device=dev)
    dx = torch.empty_like(x)
    for label, vs, ags in (("same CUs (whole chip each)", torch.cuda.current_stream(), side),
                           ("disjoint CU halves (masked streams)", half_b, half_a),
                           ("victim on half the CUs, GEMM on all", half_b, side)):
        ref, bad, worst = None, 0, 0.0
        for it in range(launches + 1):
            dlogs, dbias, dw = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(16, device=dev)
            torch.cuda.synchronize()
            if it > 0:                                   # launch 0 runs alone: the reference
                with torch.cuda.stream(ags):
                    gemm16()
            with torch.cuda.stream(vs):
                call("glowtts_actnorm_invconv_bwd", ptr(x), ptr(m), ptr(logs), ptr(bias), ptr(w), ptr(winv), ptr(dz), ptr(dld),
                     ptr(xlen), ptr(dx), ptr(dlogs), ptr(dbias), ptr(dw), B, C, T, 4)
            torch.cuda.synchronize()
            got = torch.cat([dw, dlogs, dbias])
            if ref is None:
                ref = got.clone()
            err = float((got - ref).abs().max() / ref.abs().max())
            worst = max(worst, err)
            bad += err > 5e-6
        print(f"real victim beside torch bf16 GEMM, {label:40s}: {bad:4d} / {launches} launches off, worst {worst:.2e}", flush=True)


if __name__ == "__main__":
    main()
