"""Per-workgroup timeline of k_unit_gemms_sk (BE_SK_TRACE=1 makes every workgroup stamp its start / end on the 100 MHz clock into the
tail of the training scratch): run one unit's backward of a given shape and print, per problem, the spread of the workgroups' lifetimes
against the launch's span.  usage (GPU box): BE_SK_TRACE=1 python tools/sk_trace.py [cin cout ks] ..."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
from be_hip import native, train  # noqa: E402
from be_hip.native import check, dptr, lib, stream_ptr  # noqa: E402

assert os.environ.get("BE_SK_TRACE"), "run with BE_SK_TRACE=1"
DEV = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(5)


def unit(n, hw, cin, cout, ks):
    w = (torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks) ** 0.5)).to(DEV)
    b = (torch.randn(cout, generator=g) * 0.1).to(DEV)

    class P:
        fwd = {0: native.conv_pack(w, b, bn=None)}
        dg = {}
    nd = lib().be_conv_dgrad_packed_floats(cout, cin, ks)
    dw_, db_ = train._new(nd, DEV), train._new((cin + 31) // 32 * 32, DEV)
    check(lib().be_conv_pack_dgrad_f32(dptr(w), cout, cin, ks, 0, dptr(dw_), dptr(db_), stream_ptr(DEV)), "pack dgrad")
    P.dg[0] = (dw_, db_)
    gamma, beta = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    x = torch.randn(n, hw, hw, cin, generator=g).to(DEV)
    dout = torch.randn(n, hw, hw, cout, generator=g).to(DEV)
    rm, rv = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    out, saved = train._unit_fwd(x, P, 0, cout, ks, gamma, beta, rm, rv, None, True)
    grads = (torch.empty(cout, device=DEV), torch.empty(cout, device=DEV), torch.empty_like(w), torch.empty(cout, device=DEV))
    train._Scratch.get(DEV).view(torch.int64)[-8192:].zero_()
    for _ in range(5):
        train._unit_bwd(x, dout, saved, gamma, P.dg[0], None, ks, 0, *grads)
    torch.cuda.synchronize()
    sc = train._Scratch.get(DEV)
    tail = sc.view(torch.int64)[-8192:].cpu().numpy().reshape(-1, 8)          # the last 64 KB: 1024 workgroups x 8 words
    return tail


def report(tag, tr):
    live = tr[(tr[:, 1] > tr[:, 0]) & (tr[:, 0] > 0)]
    if not len(live):
        print(tag, "no stamps")
        return
    t0, t1 = live[:, 0].min(), live[:, 1].max()
    span = (t1 - t0) / 100.0
    print(f"{tag}: {len(live)} workgroups, span {span:.1f} us (first start .. last end)")
    prob = live[:, 2] & 255
    for p in sorted(set(prob.tolist())):
        m = live[prob == p]
        st, en = (m[:, 0] - t0) / 100.0, (m[:, 1] - t0) / 100.0
        life = en - st
        segs = (m[:, 2] >> 32)
        name = {0: "wgrad a", 1: "wgrad b", 2: "conv a", 3: "conv b"}[p]
        print(f"   {name:8s} {len(m):4d} wgs  start {st.min():5.1f}..{st.max():5.1f}  end {en.min():6.1f}..{en.max():6.1f} (median {np.median(en):6.1f})"
              f"  life min/med/max {life.min():6.1f} {np.median(life):6.1f} {life.max():6.1f} us  segments/wg {segs.mean():.2f}")
        if p >= 2:
            ph = m[:, 4:8].astype(np.float64)
            print("            median cycles per workgroup: entry wait %.0f, prologue %.0f, K loop %.0f, stores %.0f; life in cycles at 2.4 GHz %.0f"
                  % (*np.median(ph, axis=0), np.median(life) * 2400))
    # how busy the chip's slots are over time: fraction of workgroups alive at 10 points of the span
    pts = np.linspace(0, span, 11)[1:-1]
    st, en = (live[:, 0] - t0) / 100.0, (live[:, 1] - t0) / 100.0
    print("   alive at 10%..90% of the span:", " ".join(f"{int(((st <= q) & (en > q)).sum()):4d}" for q in pts))
    # the shader clock each workgroup saw: its own s_memtime delta (shader cycles) over its own s_memrealtime delta (100 MHz)
    cyc = (live[:, 3] >> 8).astype(np.float64)
    ghz = cyc / ((live[:, 1] - live[:, 0]).astype(np.float64) * 10.0)
    print("   shader clock over the workgroups' lifetimes: median %.3f GHz (min %.3f, max %.3f)" % (np.median(ghz), ghz.min(), ghz.max()))
    xcc = live[:, 3] & 15
    print("   workgroups per XCC:", np.bincount(xcc.astype(int), minlength=8).tolist())


shapes = [(384, 384, 3), (256, 256, 3), (256, 384, 1)]
if len(sys.argv) > 3:
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
for cin, cout, ks in shapes:
    report(f"{cin}->{cout} {ks}x{ks} @6x6 n=64", unit(64, 6, cin, cout, ks))
