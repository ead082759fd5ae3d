"""Mutation fuzzing of the two asset parsers (csrc/assets.cpp: OBJ, Radiance RGBE) under AddressSanitizer + UBSan (host
code only).  Builds /tmp/fuzz/harness from csrc/assets.cpp with g++ -fsanitize=address,undefined and feeds it valid files
with bytes flipped, spans deleted / duplicated, truncations, digits replaced by huge numbers.  Any sanitizer report or
crash fails; a parse error is the expected outcome.  usage: python scripts/fuzz_assets.py [cases per kind, default 400]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from cuda_path_tracer_amd import generators as G
from tests.hdr_util import random_rgbe, write_hdr

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
W = "/tmp/fuzz"
os.makedirs(W + "/cases", exist_ok=True)
csrc = os.path.join(ROOT, "cuda_path_tracer_amd", "csrc")
harness_src = os.path.join(W, "harness.cpp")
open(harness_src, "w").write(r'''
#include "scene_host.h"
#include <cstdio>
#include <cstring>
using namespace mipt;
int main(int argc, char **argv) {
    int ok = 0, bad = 0;
    for (int i = 2; i < argc; i++) {
        std::string err;
        if (!strcmp(argv[1], "obj")) {
            HostMesh m;
            const float p[3] = {0, 0, 0}, r[3] = {0.1f, 0.2f, 0.3f}, s[3] = {1, 2, 3};
            if (load_obj(argv[i], p, r, s, m, err)) ok++; else bad++;
        } else {
            int w, h;
            std::vector<uint32_t> t;
            if (load_hdr_rgbe(argv[i], w, h, t, err)) ok++; else bad++;
        }
    }
    printf("ok %d rejected %d\n", ok, bad);
    return 0;
}
''')
subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-w", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + csrc, "-pthread",
                       harness_src, os.path.join(csrc, "assets.cpp"), "-o", W + "/harness"])
rng = np.random.default_rng(2024)


def mutate(data):
    b = bytearray(data)
    for _ in range(int(rng.integers(1, 6))):
        kind = int(rng.integers(0, 6))
        if not b:
            break
        i = int(rng.integers(0, len(b)))
        if kind == 0:
            b[i] = int(rng.integers(0, 256))
        elif kind == 1:
            del b[i:i + int(rng.integers(1, 64))]
        elif kind == 2:
            b[i:i] = b[i:i + int(rng.integers(1, 64))]
        elif kind == 3:
            del b[i:]
        elif kind == 4:
            b[i:i] = rng.choice([b"99999999999999999999", b"-1", b"/", b"//", b"\r", b"\x00", b"1e9999", b"nan", b" \n", b"f ", b"\x02\x02\xff\xff"])
        else:
            b[i:i + 1] = bytes([b[i] ^ (1 << int(rng.integers(0, 8)))])
    return bytes(b)


seeds = {"obj": [], "hdr": []}
G.write_icosphere_obj(W + "/s0.obj", 2, 1.5)
G.write_icosphere_obj(W + "/s1.obj", 1, 1.0, form="p/t/n", eol="\r\n")
G.write_cube_obj(W + "/s2.obj")
seeds["obj"] = [open(W + "/s%d.obj" % i, "rb").read() for i in range(3)]
write_hdr(W + "/s0.hdr", random_rgbe(9, 16, seed=1), rle=True)
write_hdr(W + "/s1.hdr", random_rgbe(5, 7, seed=2), rle=False)
write_hdr(W + "/s2.hdr", random_rgbe(3, 300, seed=3), rle=True)
seeds["hdr"] = [open(W + "/s%d.hdr" % i, "rb").read() for i in range(3)]
rc = 0
for kind in ("obj", "hdr"):
    paths = []
    for k in range(N):
        p = "%s/cases/c%d.%s" % (W, k, kind)
        open(p, "wb").write(mutate(seeds[kind][k % 3]))
        paths.append(p)
    for lo in range(0, N, 100):
        r = subprocess.run([W + "/harness", kind] + paths[lo:lo + 100], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", MIPT_BUILD_THREADS="2"))
        print(kind, lo, r.stdout.strip(), "rc", r.returncode)
        if r.returncode != 0:
            print(r.stderr[-3000:])
            rc = 1
sys.exit(rc)
