"""BASELINE configs[0]/[4] support: time the full-size DCAE (configs/DC_AE_84_pretrain.yaml) encode + decode of
240x120x84 frames on the MI355X (HIP, NHWC) and, for one frame, on the host CPU with the oracle."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.dcae import CONFIG_DCAE_84, AutoencoderDC as OracleAE
from ladcast_amd.models import AutoencoderDC

torch.manual_seed(1234)
o = OracleAE.from_config(CONFIG_DCAE_84).eval()
g = AutoencoderDC.from_config(CONFIG_DCAE_84)
g.load_state_dict(o.state_dict(), strict=True)
g = g.cuda().eval()
res = {}
for frames in (1, 8, 32):
    x = torch.randn(frames, 84, 120, 240, device="cuda"); st = torch.randn(1, 5, 120, 240, device="cuda")
    z = g.encode(x, static_conditioning_tensor=st).latent; y = g.decode(z).sample; torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n): z = g.encode(x, static_conditioning_tensor=st).latent
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): y = g.decode(z).sample
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / n
    res[f"gpu_{frames}"] = dict(encode_ms=round(te * 1e3, 2), decode_ms=round(td * 1e3, 2), encode_tflops=round(0.695 * frames / te, 1), decode_tflops=round(0.7814 * frames / td, 1))
    print(frames, res[f"gpu_{frames}"], flush=True)
x = torch.randn(1, 84, 120, 240); st = torch.randn(1, 5, 120, 240)
with torch.no_grad():
    o.encode(x, static_conditioning_tensor=st)
    t0 = time.perf_counter(); z = o.encode(x, static_conditioning_tensor=st).latent; te = time.perf_counter() - t0
    t0 = time.perf_counter(); y = o.decode(z).sample; td = time.perf_counter() - t0
res["cpu_oracle_1"] = dict(encode_ms=round(te * 1e3, 1), decode_ms=round(td * 1e3, 1), threads=torch.get_num_threads())
print(json.dumps(res))
