"""Mode resolution of a DAS plan WITHOUT a GPU: ``qups_amd/csrc/plan_modes.h`` -- the pure functions ``qdas_plan_create`` takes its decisions
from (request analysis, reciprocal / fold / mirror modes, the probe chain, workgroups per tile, frames per launch, and the launcher's own admission rules) --
compiled with g++ into ``tests/modes/enumerate_modes.cpp`` and walked over >= 12 000 descriptors x every combination of the device facts: every draw ends in a
legal launch configuration or names why not.  A second build with a known historical bug injected (the lateral-mirror mode kept on the re-basing
configuration) must FAIL: the net catches what it is there for."""
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qups_amd", "csrc")
SRC = os.path.join(ROOT, "tests", "modes", "enumerate_modes.cpp")


def _build(inc, exe):
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", inc, SRC, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_descriptor_resolves_to_a_legal_launch_or_names_why_not(tmp_path):
    exe = str(tmp_path / "enumerate_modes")
    _build(CSRC, exe)
    r = subprocess.run([exe, "12000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "mode enumeration OK" in r.stdout, r.stdout[-3000:]
    m = re.search(r"enumerated (\d+) descriptors, (\d+) walks of the geometry facts: (\d+) plans resolved", r.stdout)
    assert m and int(m.group(1)) >= 10000 and int(m.group(3)) >= 50000, r.stdout[:500]
    reached = {int(c) for c in re.findall(r" (\d+):\d+", r.stdout.split("launch configurations reached:")[1].splitlines()[0])}
    assert reached >= {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21}, reached      # (10 / 11: table-driven delays, qdas_das_lut's own path)


def test_the_enumeration_catches_an_injected_mode_bug(tmp_path):
    inc = tmp_path / "inc"
    inc.mkdir()
    shutil.copy(os.path.join(CSRC, "das_tile_cfg.h"), inc / "das_tile_cfg.h")
    src = open(os.path.join(CSRC, "plan_modes.h")).read()
    bad = src.replace("s.big = 1; s.mir = false; s.tc_sym = 0;", "s.big = 1; s.tc_sym = 0;")                 # mirror mode kept with the re-basing configuration
    assert bad != src
    bad = bad.replace('"../../include/qdas.h"', '"' + os.path.join(ROOT, "include", "qdas.h") + '"')
    (inc / "plan_modes.h").write_text(bad)
    exe = str(tmp_path / "enumerate_bad")
    _build(str(inc), exe)
    r = subprocess.run([exe, "12000"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "mirror mode on the re-basing configuration" in r.stdout, r.stdout[-2000:]


def test_no_function_of_the_api_file_is_longer_than_200_lines():
    """qdas_plan_create was one 715-line function; the split must not grow back"""
    lines = open(os.path.join(CSRC, "qdas_api.hip")).read().splitlines()
    start, name, worst = None, "", (0, "")
    for k, l in enumerate(lines):
        if start is None and l and not l[0].isspace() and l.rstrip().endswith("{") and "(" in l and not l.startswith(("struct", "namespace", "//", "#", "}")):
            start, name = k, l
        elif start is not None and l.startswith("}"):
            if k - start > worst[0]:
                worst = (k - start, name)
            start = None
    assert 0 < worst[0] <= 200, worst
