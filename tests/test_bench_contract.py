"""CPU box: the parts of bench.py that do not need a GPU -- the workload table against BASELINE.json's configurations, the algorithmic bytes per frame
(SURVEY.md 8d), the default variant list (every BASELINE configuration is in the driver's line), and the shape of the cpu_baseline object."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_workloads_cover_the_baseline_configurations():
    b = _bench()
    W = b.WORKLOADS
    assert W["c1"][:6] == (1280, 720, "yuv420p", 640, 360, "yuv420p")                  # configs[0]
    assert W["c2a"][:6] == (3840, 2160, "yuv420p", 3840, 2160, "rgb24")                # configs[1] (the metric's configuration)
    assert W["c3b"][:6] == (7680, 4320, "yuv420p10le", 3840, 2160, "p010le")           # configs[2], and its same-size twin
    assert W["c3a"][:6] == (7680, 4320, "yuv420p10le", 7680, 4320, "p010le")
    assert W["c4"][:6] == (1920, 1080, "nv12", 1920, 1080, "bgr0")                     # configs[3]
    assert W["c5"][:6] == (3840, 2160, "gbrpf32le", 3840, 2160, "yuv444p16le")         # configs[4]
    for must in ("c2b", "c3a", "c3b", "c4", "c5", "c1"):
        assert must in b.AUTO_VARIANTS, f"{must} is not in the default run's variants"
    assert b.HBM_PEAK_GBS == 8000.0


def test_algorithmic_bytes_per_frame():
    b = _bench()
    ab = lambda n: b.algorithmic_bytes(*b.WORKLOADS[n][:6])
    assert ab("c2a") == 37324800 and ab("c2b") == 37324800      # 4K: 1.5 B in + 3 B out per pixel
    assert ab("c4") == 1920 * 1080 * (1.5 + 4)
    assert ab("c3a") == 7680 * 4320 * 3 * 2                      # 8K 10-bit in 16-bit words, 4:2:0 in and out
    assert ab("c3b") == 7680 * 4320 * 3 + 3840 * 2160 * 3
    assert ab("c5") == 3840 * 2160 * (12 + 6)
    assert ab("c1") == 1280 * 720 * 1.5 + 640 * 360 * 1.5


def test_cpu_baseline_object_and_measured_ratios():
    b = _bench()
    rv = json.load(open(os.path.join(ROOT, "profiles", "ref_vs_port.json")))
    for k in ("c1", "c2a", "c2b", "c3a", "c3b", "c4", "c5"):
        assert 0.5 < rv[k]["port_over_reference"] < 1.5, (k, rv[k])      # the port stays close to the reference's C path (tools/ref_vs_port.sh)
    r = b.cpu_baseline("c2a", seconds=0.3, others=("c1",), other_seconds=0.2)
    for key in ("value", "unit", "cores", "kind", "sample", "value_1thread", "port_over_reference_1thread", "reference_c_estimate_all_threads", "configs"):
        assert key in r, key
    assert r["kind"] == "port" and r["unit"] == "Mpixels/s" and r["value"] > 0 and "c1" in r["configs"]
