"""AddressSanitizer run of the C-ABI shim's HOST side ("Race detection / sanitizers": "compile a -fsanitize=address host
build of the C-ABI shim; kernels validated by goldens").  `genpercept_amd.build.build_asan_host_library()` compiles the product sources host-only
with -fsanitize=address; a child process with clang's ASAN runtime preloaded binds it through the SAME ctypes declarations as the product
(`engine.SYMBOLS`, `GpConfig`, `GpTimings`) and drives every entry point that has host logic and does not need a GPU: argument checks, engine
creation on a box without a device (error path, the handle stays usable for host calls), tensor registration with the fp32 / fp16 / bf16
conversions, context / timestep storage, the timings struct (a ctypes mirror smaller than the C struct would be a heap overflow here: the
ADVICE r4 item about `gp_timings`), the launch-log buffer with tight capacities, the size helpers.  No compute, no GPU; any ASAN report
fails the test.  PYTHONMALLOC=malloc: ctypes buffers then come from the
sanitizer's own allocator (red zones) instead of pymalloc arenas."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, {root!r})
from genpercept_amd import engine as ge
lib = C.CDLL({lib!r})
for name, (res, args) in ge.SYMBOLS.items():   # the product's own declarations: every header symbol must bind in the sanitizer build too
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
assert lib.gp_version().decode()
assert lib.gp_element_dtype() == 2   # GP_DT_BF16: the sanitizer build is the bf16 element type
cfg = ge.GpConfig()
lib.gp_default_config(C.byref(cfg))
assert list(cfg.unet_block_out) == [320, 640, 1280, 1280] and abs(cfg.vae_scaling_factor - 0.18215) < 1e-7
# size helpers (host arithmetic)
h, w = C.c_int(0), C.c_int(0)
lib.gp_resize_max_res_size(3024, 4032, 768, C.byref(h), C.byref(w))
assert (h.value, w.value) == (576, 768)
assert lib.gp_latent_size(768) == 96 and lib.gp_dpt_out_size(96) == 768 and lib.gp_packed_rows(320) % 256 == 0
# argument checks
assert lib.gp_create(None, None) != 0
assert lib.gp_get_timings(None, None) != 0 and lib.gp_reset_timings(None) != 0 and lib.gp_get_launch_log(None, None, 0) == -1
hdl = C.c_void_p()
cfg.device = 0
st = lib.gp_create(C.byref(cfg), C.byref(hdl))
has_gpu = st == 0
assert hdl.value, "gp_create hands the engine out even when it reports an error (so that gp_last_error can be read)"
if not has_gpu:
    assert lib.gp_last_error(hdl).decode() != ""
# tensor registration: fp32 copy, fp16 / bf16 conversion loops, bad dtype, NULL arguments, a 0-d tensor
rng = np.random.default_rng(0)
shape = (C.c_int64 * 4)(8, 4, 3, 3)
x32 = rng.standard_normal((8, 4, 3, 3)).astype(np.float32)
assert lib.gp_load_tensor(hdl, b"a.weight", x32.ctypes.data, shape, 4, 0) == 0
x16 = x32.astype(np.float16)
assert lib.gp_load_tensor(hdl, b"b.weight", x16.ctypes.data, shape, 4, 1) == 0
xbf = (x32.view(np.uint32) >> 16).astype(np.uint16)
assert lib.gp_load_tensor(hdl, b"c.weight", xbf.ctypes.data, shape, 4, 2) == 0
assert lib.gp_load_tensor(hdl, b"a.weight", x32.ctypes.data, shape, 4, 0) == 0      # overwrite an existing name
assert lib.gp_load_tensor(hdl, b"d", x32.ctypes.data, shape, 4, 99) != 0 and b"dtype" in lib.gp_last_error(hdl)
assert lib.gp_load_tensor(hdl, None, x32.ctypes.data, shape, 4, 0) != 0 and lib.gp_load_tensor(hdl, b"e", None, shape, 4, 0) != 0
one = np.ones(1, np.float32)
assert lib.gp_load_tensor(hdl, b"scalar", one.ctypes.data, None, 0, 0) == 0
ctx = rng.standard_normal((2, 1024)).astype(np.float32)
assert lib.gp_set_context(hdl, ctx.ctypes.data, 2, 1024) == 0
assert lib.gp_set_context(hdl, None, 2, 1024) != 0 and lib.gp_set_context(hdl, ctx.ctypes.data, 0, 1024) != 0
assert lib.gp_set_timestep(hdl, 1.0) == 0
assert lib.gp_set_profile(hdl, 1) == 0
# the timings struct is written whole into a heap object of exactly ctypes' size
tm = ge.GpTimings()
assert lib.gp_reset_timings(hdl) == 0
assert lib.gp_get_timings(hdl, C.byref(tm)) == 0 and tm.n_launches == 0
# launch log: query the size, then read into exactly-sized and into too-small buffers
need = lib.gp_get_launch_log(hdl, None, 0)
assert need >= 1
for cap in (need, 1, 2):
    buf = C.create_string_buffer(cap)
    assert lib.gp_get_launch_log(hdl, buf, cap) == need
if not has_gpu:   # finalize needs the device: the error path must leave the engine destroyable
    assert lib.gp_finalize(hdl) != 0 and lib.gp_last_error(hdl).decode() != ""
fl = C.c_double(-1.0)
assert lib.gp_halo_executed_flops(hdl, C.byref(fl)) == 0 and fl.value == 0.0 and lib.gp_halo_executed_flops(None, C.byref(fl)) != 0
assert lib.gp_pack_weight_phases(None, 8, 4, 64, None) != 0 and lib.gp_pack_weight_phases(x32.ctypes.data, 8, 4, 3, x32.ctypes.data) != 0   # (bad cin_pad)
assert lib.gp_conv2d_up2(None, None, None, None, None, None, 1, 16, 16, 64, 64, None) != 0
# r6 entry points: ABI version, engine precision (state rules: before gp_finalize only; bad values refused), the phase conv with statistics, weight-count checks
assert lib.gp_abi_version() >= 3
assert lib.gp_get_precision(hdl) == 0 and lib.gp_set_precision(hdl, 1) == 0 and lib.gp_get_precision(hdl) == 1
assert lib.gp_set_precision(hdl, 7) != 0 and lib.gp_set_precision(None, 1) != 0 and lib.gp_set_precision(hdl, 0) == 0 and lib.gp_get_precision(hdl) == 0
assert lib.gp_conv2d_up2_stats(None, None, None, None, None, None, 1, 16, 16, 64, 64, None, None, 32, 1e-6, None, None, None) != 0
assert lib.gp_pack_weight_phases(x32.ctypes.data, -1, 4, 64, x32.ctypes.data) != 0 and lib.gp_pack_weight_phases(x32.ctypes.data, 8, 0, 64, x32.ctypes.data) != 0
assert lib.gp_pack_weight(x32.ctypes.data, 0, 4, 3, 64, 0, x32.ctypes.data) != 0
assert lib.gp_flash_attention_split(None, 192, None, 1, 64, 1, None) != 0 and lib.gp_flash_attention_split(x32.ctypes.data, 100, x32.ctypes.data, 1, 64, 1, None) != 0   # (ld < 3 C)
ev = C.c_longlong(-1)
lib.gp_saturation_events(hdl, C.byref(ev), 1)
lib.gp_destroy(hdl)
lib.gp_destroy(None)
print("ASAN_HOST_OK", flush=True)
"""


def test_c_abi_host_paths_under_address_sanitizer(tmp_path):
    from genpercept_amd import build as gb
    try:
        lib = gb.build_asan_host_library()
        rt = gb.asan_runtime()
    except RuntimeError as e:  # (no hipcc / no clang runtime on this machine)
        pytest.skip(str(e))
    script = tmp_path / "asan_worker.py"
    script.write_text(_WORKER.format(root=ROOT, lib=lib))
    env = dict(os.environ, LD_PRELOAD=rt, PYTHONMALLOC="malloc", ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=86:protect_shadow_gap=0",
               HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", ""))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "ERROR: " not in out, out[-4000:]
    assert r.returncode == 0 and "ASAN_HOST_OK" in r.stdout, out[-4000:]


_CANARY = r"""
import ctypes as C, sys
sys.path.insert(0, {root!r})
from genpercept_amd import engine as ge
lib = C.CDLL({lib!r})
for name, (res, args) in ge.SYMBOLS.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
cfg = ge.GpConfig()
lib.gp_default_config(C.byref(cfg))
hdl = C.c_void_p()
lib.gp_create(C.byref(cfg), C.byref(hdl))
small = C.create_string_buffer(C.sizeof(ge.GpTimings) - 8)   # what a caller compiled against a SHORTER gp_timings would pass
lib.gp_get_timings.argtypes = [C.c_void_p, C.c_void_p]
lib.gp_get_timings(hdl, small)
print("NOT_CAUGHT", flush=True)
"""


def test_the_sanitizer_build_catches_a_short_timings_struct(tmp_path):
    """the harness has teeth: a heap buffer 8 bytes shorter than `gp_timings` (the ABI break ADVICE r4 described: a field appended to a public
    struct) is reported by the sanitizer as a heap-buffer-overflow inside gp_get_timings"""
    from genpercept_amd import build as gb
    try:
        lib = gb.build_asan_host_library()
        rt = gb.asan_runtime()
    except RuntimeError as e:
        pytest.skip(str(e))
    script = tmp_path / "asan_canary.py"
    script.write_text(_CANARY.format(root=ROOT, lib=lib))
    env = dict(os.environ, LD_PRELOAD=rt, PYTHONMALLOC="malloc", ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=86:protect_shadow_gap=0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 86 and "heap-buffer-overflow" in out and "NOT_CAUGHT" not in out, out[-3000:]
