"""Board power / shader clock telemetry through a sustained loop of the model's GEMM launches.

A sampler thread reads the amdgpu hwmon / sysfs nodes (power1_average|power1_input, power1_cap, freq1_input, pp_dpm_sclk's
active level; no root needed) every PERIOD_MS (default 10 ms) while the main thread replays one launch shape back to back for
SECONDS (default 10) per shape; nodes that do not exist on the box fall back to one `amd-smi metric` / `rocm-smi` call per ~100 ms.
Prints one JSON object: per shape the time per launch and TFLOP/s over the whole window, and the power / clock trace reduced to
(min, median, mean, max) plus the raw samples of the first and the last second.  Measurement aid: not imported by the product.

usage: python tools/power_trace.py [bf16x3|bf16|vendor_bf16|zeros]   (zeros = bf16x3 on all-zero operands: what the clock does without data toggling)
"""
import glob, json, os, subprocess, sys, threading, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

MODE = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
SECONDS = float(os.environ.get("SECONDS_PER_SHAPE", "10"))
PERIOD = float(os.environ.get("PERIOD_MS", "10")) / 1e3

# the five launch shapes that carry the 375M forward's GEMM time at one member (grouped problems of one launch)
SHAPES = {
    "dual qkv": [(1800, 4608, 1536), (450, 4608, 1536)], "dual out": [(1800, 1536, 1536), (450, 1536, 1536)],
    "dual ff up": [(1800, 6144, 1536), (450, 6144, 1536)], "dual ff down": [(1800, 1536, 6144), (450, 1536, 6144)],
    "single qkv+mlp": [(2250, 6144, 1536), (2250, 4608, 1536)], "single out": [(2250, 1536, 7680)],
}


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.nodes = {}
        # the box is a slice of a multi-GPU node: every card is in sysfs, only one is ours -> match the PCI address of HIP device 0
        want = None
        try:
            pr = torch.cuda.get_device_properties(0)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}".lower()
        except Exception:
            pass
        self.pci = want
        for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            if want is not None and want not in os.path.realpath(card).lower():
                continue
            hw = sorted(glob.glob(card + "/hwmon/hwmon*"))
            if not hw:
                continue
            self.card = card
            for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "temp1_input"):
                p = os.path.join(hw[0], name)
                if _read(p) is not None:
                    self.nodes[name] = p
            if _read(card + "/pp_dpm_sclk") is not None:
                self.nodes["pp_dpm_sclk"] = card + "/pp_dpm_sclk"
            break
        self.samples, self.stop, self.lock = [], False, threading.Lock()
        self.cli = None
        if not any(k.startswith("power1_a") or k == "power1_input" for k in self.nodes):
            for cmd in (["amd-smi", "metric", "-p", "-c", "--json"], ["rocm-smi", "--showpower", "--showclocks", "--json"]):
                try:
                    subprocess.run(cmd, capture_output=True, timeout=10, check=True)
                    self.cli = cmd
                    break
                except Exception:
                    pass

    def one(self):
        s = {"t": time.perf_counter()}
        for k, p in self.nodes.items():
            v = _read(p)
            if v is None:
                continue
            if k == "pp_dpm_sclk":
                act = [ln for ln in v.splitlines() if ln.rstrip().endswith("*")]
                if act:
                    s["sclk_mhz"] = float(act[0].split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
            elif k.startswith("power1"):
                s[k + "_w"] = float(v) / 1e6
            elif k == "freq1_input":
                s["freq1_mhz"] = float(v) / 1e6
            else:
                s[k] = float(v)
        if self.cli is not None:
            try:
                s["cli"] = subprocess.run(self.cli, capture_output=True, timeout=5, text=True).stdout[:2000]
            except Exception:
                pass
        return s

    def run(self):
        while not self.stop:
            s = self.one()
            with self.lock:
                self.samples.append(s)
            time.sleep(PERIOD if self.cli is None else 0.1)

    def take(self):
        with self.lock:
            out, self.samples = self.samples, []
        return out


def stats(vals):
    if not vals:
        return None
    v = sorted(vals)
    return dict(min=round(v[0], 2), median=round(v[len(v) // 2], 2), mean=round(sum(v) / len(v), 2), max=round(v[-1], 2), n=len(v))


def main():
    import ladcast_amd.hip as hip

    smp = Sampler()
    res = {"mode": MODE, "pci": smp.pci, "card": getattr(smp, "card", None), "seconds_per_shape": SECONDS, "period_ms": PERIOD * 1e3, "nodes": smp.nodes, "cli": smp.cli, "device": torch.cuda.get_device_name(0), "shapes": {}}
    smp.start()
    time.sleep(1.0)
    idle = smp.take()
    res["idle"] = {k: stats([s[k] for s in idle if k in s]) for k in ("power1_average_w", "power1_input_w", "sclk_mhz", "freq1_mhz", "power1_cap_w")}
    for name, probs in SHAPES.items():
        flops = sum(2.0 * M * N * K for M, N, K in probs)
        if MODE == "vendor_bf16":
            ops = [(torch.randn(M, K, device="cuda").bfloat16(), torch.randn(N, K, device="cuda").bfloat16().t()) for M, N, K in probs]
            fn = lambda: [torch.matmul(a, w) for a, w in ops]  # noqa: E731
        else:
            ps = []
            for M, N, K in probs:
                mk = torch.zeros if MODE == "zeros" else torch.randn
                A, W, C = mk(M, K, device="cuda"), mk(N, K, device="cuda"), torch.empty(M, N, device="cuda")
                fl = hip.GEMM_A_SPLIT | (hip.GEMM_BF16_1TERM if MODE == "bf16" else 0)
                if MODE == "bf16":  # plain bf16 operand rows
                    Ab = torch.zeros_like(A)
                    Ab.view(torch.bfloat16)[:, :K] = A.bfloat16()
                    ps.append(hip.gemm_problem(Ab, hip.pack_weight_bf16(W), C, M=M, N=N, K=K, flags=fl))
                else:
                    ps.append(hip.gemm_problem(hip.pack_weight_bf16x2(A), hip.pack_weight_bf16x2(W), C, M=M, N=N, K=K, flags=fl))
            fn = lambda: hip.gemm_grouped(ps, split_bf16=True)  # noqa: E731
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        smp.take()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < SECONDS:
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
            n += 200
        dt = time.perf_counter() - t0
        ss = smp.take()
        ent = dict(launches=n, us_per_launch=round(1e6 * dt / n, 2), tflops=round(flops * n / dt / 1e12, 1))
        for k in ("power1_average_w", "power1_input_w", "sclk_mhz", "freq1_mhz", "temp1_input"):
            st = stats([s[k] for s in ss if k in s])
            if st:
                ent[k] = st
        key = "power1_average_w" if any("power1_average_w" in s for s in ss) else "power1_input_w"
        t_first = ss[0]["t"] if ss else 0
        ent["first_second"] = [(round(s["t"] - t_first, 3), s.get(key), s.get("sclk_mhz", s.get("freq1_mhz"))) for s in ss if s["t"] - t_first < 1.0][::5]
        ent["last_second"] = [(round(s["t"] - t_first, 3), s.get(key), s.get("sclk_mhz", s.get("freq1_mhz"))) for s in ss if ss[-1]["t"] - s["t"] < 1.0][::5]
        if smp.cli and ss:
            ent["cli_sample"] = ss[len(ss) // 2].get("cli")
        res["shapes"][name] = ent
        time.sleep(0.5)
    smp.stop = True
    print(json.dumps(res))


if __name__ == "__main__":
    main()
