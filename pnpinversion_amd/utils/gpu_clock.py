"""Shader clock / socket power / temperature of the GPU a benchmark runs on, sampled from a side thread (bench.py's `clock` object:
VERDICT r5 item 4 -- a 5 % swing between boxes or rounds must be attributable from the record).  Host plumbing only; no reference
counterpart.  Sources, first that answers: amdgpu's sysfs hwmon files of the PCI device torch reports (no privileges, no subprocess),
then the amdsmi Python bindings.  Nothing here may raise into the caller: a box without either source yields {"source": None}."""
import glob
import os
import threading
import time


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return None


def _sysfs_device_dir(device_index):
    """/sys/bus/pci/devices/<domain:bus:dev.fn> of the torch device, or the only amdgpu card of the box"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        d = os.path.join("/sys/bus/pci/devices", bdf)
        if os.path.isdir(d):
            return d
    except Exception:
        pass
    cards = []
    for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        dev = os.path.join(c, "device")
        try:
            if open(os.path.join(dev, "vendor")).read().strip() == "0x1002":
                cards.append(os.path.realpath(dev))
        except Exception:
            continue
    return cards[0] if len(cards) == 1 else None


class _SysfsSource:
    name = "sysfs-hwmon"

    def __init__(self, device_index):
        d = _sysfs_device_dir(device_index)
        self.files = {}
        if d is None:
            return
        for hw in sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*"))):
            for key, names in (("sclk_mhz", ("freq1_input",)), ("power_w", ("power1_average", "power1_input")),
                               ("temp_c", ("temp1_input",)), ("temp_junction_c", ("temp2_input",))):
                for n in names:
                    p = os.path.join(hw, n)
                    if key not in self.files and _read_int(p) is not None:
                        self.files[key] = p
        self.device = d

    def ok(self):
        return "sclk_mhz" in self.files or "power_w" in self.files

    def sample(self):
        out = {}
        for key, p in self.files.items():
            v = _read_int(p)
            if v is None:
                continue
            out[key] = v / 1e6 if key in ("sclk_mhz", "power_w") else v / 1e3      # Hz -> MHz, uW -> W, m degC -> degC
        return out


class _AmdSmiSource:
    name = "amdsmi"

    def __init__(self, device_index):
        self.h = None
        try:
            import amdsmi
            self.smi = amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            want = None
            try:
                import torch
                p = torch.cuda.get_device_properties(device_index)
                want = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            except Exception:
                pass
            for h in hs:
                try:
                    if want is not None and amdsmi.amdsmi_get_gpu_device_bdf(h).lower() == want:
                        self.h = h
                except Exception:
                    continue
            if self.h is None and len(hs) == 1:
                self.h = hs[0]
        except Exception:
            self.h = None

    def ok(self):
        return self.h is not None and bool(self.sample())

    def sample(self):
        out = {}
        s = self.smi
        try:
            c = s.amdsmi_get_clock_info(self.h, s.AmdSmiClkType.GFX)
            v = c.get("clk", c.get("cur_clk"))
            if isinstance(v, (int, float)):
                out["sclk_mhz"] = float(v)
        except Exception:
            pass
        try:
            p = s.amdsmi_get_power_info(self.h)
            v = p.get("current_socket_power", p.get("average_socket_power"))
            if isinstance(v, (int, float)):
                out["power_w"] = float(v)
        except Exception:
            pass
        try:
            v = s.amdsmi_get_temp_metric(self.h, s.AmdSmiTemperatureType.HOTSPOT, s.AmdSmiTemperatureMetric.CURRENT)
            if isinstance(v, (int, float)):
                out["temp_junction_c"] = float(v)
        except Exception:
            pass
        return out


class ClockSampler:
    """with ClockSampler(device_index) as cs: <timed region>;  cs.summary() -> {"sclk_mhz": {"mean", "min", "max"}, ..., "samples", "source"}"""

    def __init__(self, device_index=0, period_s=0.05):
        self.period = period_s
        self.src = None
        for cls in (_SysfsSource, _AmdSmiSource):
            try:
                s = cls(device_index)
                if s.ok():
                    self.src = s
                    break
            except Exception:
                continue
        self.rows = []
        self._stop = threading.Event()
        self._th = None

    def _run(self):
        while not self._stop.is_set():
            try:
                r = self.src.sample()
                if r:
                    self.rows.append(r)
            except Exception:
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        if self.src is not None:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=2.0)
        return False

    def summary(self):
        out = {"source": self.src.name if self.src is not None else None, "samples": len(self.rows), "period_s": self.period}
        for key in ("sclk_mhz", "power_w", "temp_c", "temp_junction_c"):
            v = [r[key] for r in self.rows if key in r]
            if v:
                out[key] = {"mean": round(sum(v) / len(v), 1), "min": round(min(v), 1), "max": round(max(v), 1)}
        return out


def probe_once(device_index=0):
    """one sample outside any timed region (what the idle part reports)"""
    cs = ClockSampler(device_index)
    if cs.src is None:
        return {"source": None}
    r = {}
    try:
        r = cs.src.sample()
    except Exception:
        pass
    r["source"] = cs.src.name
    return r


if __name__ == "__main__":
    with ClockSampler(0) as c:
        time.sleep(0.5)
    print(c.summary())
