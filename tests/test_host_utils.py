"""Host-side plumbing added in round 6, CPU only (no compute calls): the clock sampler of bench.py's `clock` object on a fake amdgpu hwmon
tree, the pnpi_recon_desc binding (struct_size, layout of include/pnpi.h), and the product library's refusal of ablation-only tuning values."""
import ctypes as C
import os
import time

from pnpinversion_amd import _capi
from pnpinversion_amd.utils import gpu_clock


def test_clock_sampler_reads_amdgpu_hwmon_units(tmp_path, monkeypatch):
    hw = tmp_path / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (hw / "freq1_input").write_text("2100000000\n")          # Hz
    (hw / "power1_average").write_text("1050000000\n")        # microwatts
    (hw / "temp1_input").write_text("51000\n")                # millidegrees C (edge)
    (hw / "temp2_input").write_text("63000\n")                # junction
    monkeypatch.setattr(gpu_clock, "_sysfs_device_dir", lambda idx: str(tmp_path))
    with gpu_clock.ClockSampler(0, period_s=0.01) as cs:
        time.sleep(0.08)
        (hw / "freq1_input").write_text("1900000000\n")
        time.sleep(0.08)
    s = cs.summary()
    assert s["source"] == "sysfs-hwmon" and s["samples"] >= 4
    assert s["sclk_mhz"]["max"] == 2100.0 and s["sclk_mhz"]["min"] == 1900.0 and 1900.0 < s["sclk_mhz"]["mean"] < 2100.0
    assert s["power_w"]["mean"] == 1050.0 and s["temp_c"]["mean"] == 51.0 and s["temp_junction_c"]["max"] == 63.0
    one = gpu_clock.probe_once(0)
    assert one["source"] == "sysfs-hwmon" and one["sclk_mhz"] == 1900.0


def test_clock_sampler_without_any_source_is_inert(monkeypatch):
    monkeypatch.setattr(gpu_clock, "_sysfs_device_dir", lambda idx: None)

    class NoSmi:
        def __init__(self, idx):
            pass

        def ok(self):
            return False

    monkeypatch.setattr(gpu_clock, "_AmdSmiSource", NoSmi)
    with gpu_clock.ClockSampler(0) as cs:
        pass
    assert cs.summary() == {"source": None, "samples": 0, "period_s": 0.05}
    assert gpu_clock.probe_once(0) == {"source": None}


def test_recon_desc_binding_matches_the_header_layout():
    d = _capi.ReconDesc.make(None, 0.5, -600, 1, None)
    assert d.struct_size == C.sizeof(_capi.ReconDesc) == 40                     # uint32 + pad | ptr | float int int + pad | ptr  (LP64)
    assert _capi.ReconDesc.struct_size.offset == 0 and _capi.ReconDesc.ref_image.offset == 8
    assert _capi.ReconDesc.recon_lr.offset == 16 and _capi.ReconDesc.recon_t.offset == 20 and _capi.ReconDesc.dilate_mask.offset == 24
    assert _capi.ReconDesc.inv_x_stars.offset == 32
    assert d.recon_lr == 0.5 and d.recon_t == -600 and d.dilate_mask == 1 and not d.ref_image and not d.inv_x_stars
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pnpi.h")).read()
    body = hdr[hdr.index("uint32_t struct_size;"):hdr.index("} pnpi_recon_desc;")]
    order = [body.index(f) for f in ("struct_size;", "ref_image;", "recon_lr;", "recon_t;", "dilate_mask;", "inv_x_stars;")]
    assert order == sorted(order)                                                # the ctypes field order is the header's


def test_product_library_rejects_ablation_only_tuning_values():
    """the selectors of kernel instances that exist only in a `build --ablations` library used to be accepted and silently timed the
    default kernel; pnpi_set_tuning answers them with an error in the product build (no GPU needed: process-global knobs)."""
    lib = _capi.load_library()
    for key, bad in ((b"igemm_vpp", 1), (b"igemm_vpp", 4), (b"igemm_sched", 1), (b"igemm_sched", 2), (b"igemm_v128", 11), (b"igemm_v128", 12),
                     (b"igemm_v128", 15), (b"igemm_v320", 11), (b"igemm_v320", 12), (b"attn_pipe", 1), (b"attn_pipe", 2)):
        assert lib.pnpi_set_tuning(key, bad) != 0, (key, bad)
    for key, ok in ((b"igemm_vpp", 0), (b"igemm_sched", 0), (b"igemm_v128", 2), (b"igemm_v320", 1), (b"gn_slab", 0), (b"attn_pipe", 0)):     # the defaults are accepted
        assert lib.pnpi_set_tuning(key, ok) == 0, (key, ok)
    assert lib.pnpi_set_tuning(b"no_such_knob", 1) != 0
