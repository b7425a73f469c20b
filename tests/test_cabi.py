"""The C-ABI library loads and exports every symbol include/whisper_amd.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

from whisper_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "whisper_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/whisper_amd.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "whisper_amd/_lib.py prototypes must mirror the header one-to-one"


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/whisper_amd.h is what another host language binds: it must compile as C99 with no C++ or HIP types, and a C
    program linked against the library must reach the entry points (argument errors only -- no GPU here)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "cabi.c"
    src.write_text(
        '#include "whisper_amd.h"\n#include <stdio.h>\n#include <string.h>\n'
        'int main(void) {\n'
        '  wx_config cfg; memset(&cfg, 0, sizeof cfg);\n'
        '  wx_engine* e = 0;\n'
        '  if (!strstr(wx_version(), "gfx950")) return 1;\n'
        '  if (wx_create(0, &e) >= 0) return 2;\n'
        '  if (wx_create(&cfg, &e) >= 0) return 3;          /* all-zero dims are rejected before any HIP call */\n'
        '  if (wx_finalize(0) >= 0) return 4;\n'
        '  if (sizeof(wx_config) != 24 * 4 || sizeof(wx_decode_opts) != 9 * 4) return 5;\n'
        '  printf("%s\\n", wx_last_error());\n  return 0;\n}\n')
    exe = tmp_path / "cabi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                    "-L", libdir, "-lwhisper_amd", "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


def test_struct_layouts_match_header():
    src = open(os.path.join(ROOT, "include", "whisper_amd.h")).read()
    cfg = re.search(r"typedef struct \{([^}]*)\} wx_config;", src).group(1)
    cfg = re.sub(r"/\*.*?\*/", "", cfg, flags=re.S)
    fields = [f.strip() for decl in re.findall(r"int32_t\s+([^;]+);", cfg) for f in decl.split(",")]
    assert fields == [n for n, _ in _lib.WxConfig._fields_]
    opts = re.search(r"typedef struct \{([^}]*)\} wx_decode_opts;", src).group(1)
    opts = re.sub(r"/\*.*?\*/", "", opts, flags=re.S)
    ofields = [f.strip() for decl in re.findall(r"int32_t\s+([^;]+);", opts) for f in decl.split(",")]
    assert ofields == [n for n, _ in _lib.WxDecodeOpts._fields_]


def test_version_and_argument_errors_without_gpu():
    lib = _lib.load()
    assert b"gfx950" in lib.wx_version()
    assert lib.wx_create(None, None) < 0
    assert b"null" in lib.wx_last_error()
    bad = _lib.WxConfig(n_audio_state=100, n_text_state=100, n_audio_head=1, n_text_head=1)
    h = ctypes.c_void_p()
    assert lib.wx_create(ctypes.byref(bad), ctypes.byref(h)) == -1      # WX_ERR_INVALID before any HIP call
    assert lib.wx_finalize(None) < 0 and lib.wx_weight_bytes(None) == 0


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "whisper_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f"{f} imports the oracle"


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from whisper_amd import checkpoint as ck
    from whisper_amd.engine import WhisperEngine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        WhisperEngine(ck.dims_for("micro"), {}, precision="fp32")
