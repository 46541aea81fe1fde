"""CPU study: how much accuracy does a bf16x3 ("hi*hi + hi*lo + lo*hi", fp32 accumulate) GEMM give up
against exact fp32 on this model, per forward and over a full Heun sampler chunk?  (development aid)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle.ar_model import CONFIG_375M
from oracle import pipelines as OP
from oracle.scheduler import EDMDPMSolverMultistepScheduler
from tests.synth import make_ar, synth_known, tiny_ar_config, rel_l2

MODE = {"terms": 0}
_orig_linear = F.linear

def split(x):
    hi = x.bfloat16().float()
    lo = (x - hi).bfloat16().float()
    return hi, lo

def split3(x):
    a = x.bfloat16().float(); r = x - a
    b = r.bfloat16().float(); c = (r - b).bfloat16().float()
    return a, b, c

def linear_split(x, w, b=None):
    t = MODE["terms"]
    if t == 0 or x.shape[-1] < 64:
        return _orig_linear(x, w, b)
    if t == 3:
        xh, xl = split(x); wh, wl = split(w)
        y = _orig_linear(xh, wh) + _orig_linear(xh, wl) + _orig_linear(xl, wh)
    elif t == 6:
        x1, x2, x3 = split3(x); w1, w2, w3 = split3(w)
        y = _orig_linear(x1, w1) + (_orig_linear(x1, w2) + _orig_linear(x2, w1)) + (_orig_linear(x2, w2) + _orig_linear(x1, w3) + _orig_linear(x3, w1))
    elif t == 1:
        y = _orig_linear(x.bfloat16().float(), w.bfloat16().float())
    if b is not None:
        y = y + b
    return y

_orig_sdpa = F.scaled_dot_product_attention
ATT = {"split": False}

def sdpa_split(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
    if not ATT["split"]:
        return _orig_sdpa(q, k, v, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal)
    qs = q * (q.shape[-1] ** -0.5)
    qh, ql = split(qs); kh, kl = split(k); vh, vl = split(v)
    s_ = qh @ kh.transpose(-1, -2) + qh @ kl.transpose(-1, -2) + ql @ kh.transpose(-1, -2)
    pu = torch.exp(s_ - s_.amax(dim=-1, keepdim=True))
    ph, pl = split(pu)
    o = ph @ vh + ph @ vl + pl @ vh
    return o / pu.sum(dim=-1, keepdim=True)

F.scaled_dot_product_attention = sdpa_split
F.linear = linear_split
torch.nn.functional.linear = linear_split

def run(cfg, steps, members=1, R=4):
    m = make_ar(cfg)
    known, ts = synth_known(1), torch.tensor([2018010100])
    x = torch.randn(members, 84, R, 15, 30, generator=torch.Generator().manual_seed(3))
    res = {}
    for terms in (0, 3, 6, 1, 33):
        MODE["terms"] = 3 if terms == 33 else terms
        ATT["split"] = terms == 33
        with torch.no_grad():
            t0 = time.time()
            f = m(x, torch.tensor([0.3]), known.expand(members, -1, -1, -1, -1), time_elapsed=ts).sample
            pipe = OP.AutoRegressive2DPipeline(m, EDMDPMSolverMultistepScheduler())
            s = OP.ensemble_AR_sampler(pipe, members, R, steps, known_latents=known, timestamps=ts, sampler_type="edm") if steps else None
            res[terms] = (f, s, time.time() - t0)
    for terms in (3, 33, 6, 1):
        f, s, dt = res[terms]
        msg = f"terms={terms}{' (GEMM+attention split)' if terms == 33 else ''}: forward rel-L2 {rel_l2(f, res[0][0]):.3e}"
        if s is not None:
            msg += f" | {steps}-step Heun chunk rel-L2 {rel_l2(s, res[0][1]):.3e}"
        print(msg, f"({dt:.0f}s)", flush=True)

if __name__ == "__main__":
    print("tiny (D=256, 1+1+1 blocks), 20 steps")
    run(tiny_ar_config(heads=2, layers=1, single=1, refiner=1), 20)
    print("medium (D=512, 2+2+1 blocks), 20 steps")
    run(tiny_ar_config(heads=4, layers=2, single=2, refiner=1), 20)
    if len(sys.argv) > 1:
        print("375M, forward only + 3-step chunk")
        run(dict(CONFIG_375M), 3)
