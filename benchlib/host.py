"""What the box was doing while a leg ran: GPU clock / power / temperature from sysfs, and the host CPU description printed with cpu_baseline."""
import os

import torch


def gpu_state(dev_index):
    """Clock / power / temperature of this rank's GPU from sysfs (hwmon of the device's PCI function; readable as an ordinary user):
    sampled right before and right after the timed region so that two runs whose `value` differs can be told apart by what the
    chip was doing (DVFS: MI355X_MICROARCH.md, 'DVFS give-back' - devices differ by up to 12 % on MFMA-dense loops).  The read
    happens outside the timed region; a missing file leaves its field out."""
    import glob

    out = {}
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        out["pci_bus_id"] = bdf
        out["name"] = pr.name
        out["gcn_arch"] = getattr(pr, "gcnArchName", None)
        out["uuid"] = str(getattr(pr, "uuid", ""))
        base = f"/sys/bus/pci/devices/{bdf}"

        def rd(path):
            try:
                with open(path) as f:
                    return f.read().strip()
            except OSError:
                return None

        for hw in glob.glob(base + "/hwmon/hwmon*"):
            for key, fn, scale in (("sclk_mhz", "freq1_input", 1e-6), ("mclk_mhz", "freq2_input", 1e-6), ("power_w", "power1_input", 1e-6),
                                   ("power_w", "power1_average", 1e-6), ("power_cap_w", "power1_cap", 1e-6), ("temp_junction_c", "temp2_input", 1e-3),
                                   ("temp_edge_c", "temp1_input", 1e-3), ("temp_mem_c", "temp3_input", 1e-3)):
                v = rd(f"{hw}/{fn}")
                if v is not None and key not in out:
                    try:
                        out[key] = round(float(v) * scale, 1)
                    except ValueError:
                        pass
        lv = rd(base + "/pp_dpm_sclk")
        if lv:
            cur = [ln for ln in lv.splitlines() if ln.strip().endswith("*")]
            out["pp_dpm_sclk"] = cur[0].strip() if cur else None
    except Exception as e:  # diagnostics only: never fail the bench over it
        out["error"] = repr(e)
    return out


class GpuStateSampler:
    """Background thread: `gpu_state` every `period` seconds while the timed region runs (a sysfs read on the host; the launching thread
    is asleep inside graph replays / the final synchronise, so the GPU never waits for it).  `summary()`: min / median / max of clock and
    power, max temperature, number of samples."""

    def __init__(self, dev_index, period=0.05):
        import threading

        self.dev_index, self.period, self.samples = dev_index, period, []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(gpu_state(self.dev_index))
            self._stop.wait(self.period)

    def start(self):
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        self._thread.join(timeout=2.0)
        return self.summary()

    def summary(self):
        import statistics

        out = {"samples": len(self.samples)}
        for key in ("sclk_mhz", "power_w"):
            v = [x[key] for x in self.samples if key in x]
            if v:
                out[key] = dict(min=min(v), median=round(statistics.median(v), 1), max=max(v))
        for key in ("temp_junction_c", "temp_mem_c", "temp_edge_c"):
            v = [x[key] for x in self.samples if key in x]
            if v:
                out[key + "_max"] = max(v)
        return out


def host_description():
    """CPU model string, sockets x physical cores, threads (lscpu): printed with cpu_baseline."""
    import subprocess

    d = {"threads_online": os.cpu_count()}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {}
        for ln in txt.splitlines():
            if ":" in ln:
                k, v = ln.split(":", 1)
                kv[k.strip()] = v.strip()
        d["model"] = kv.get("Model name")
        sockets, cps = int(kv.get("Socket(s)", "0") or 0), int(kv.get("Core(s) per socket", "0") or 0)
        d["sockets"], d["physical_cores"] = sockets, sockets * cps
        d["threads_per_core"] = int(kv.get("Thread(s) per core", "0") or 0)
    except Exception as e:
        d["error"] = repr(e)
    return d

