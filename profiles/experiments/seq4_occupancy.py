"""Round 4: the in-workgroup split kernel on 128-image tiles (conv4 / conv5 of the metric step) holds 88-100 VGPRs + 64 accumulation
registers = three workgroups per CU.  Variant libraries (-DPCONV_SEQ4=1: staging loads up front, =2: interleaved) force four
(amdgpu_waves_per_eu(4, 4): 13 / 29 registers spilled around the range folds; the variant kernel lived in pconv_gemm.hip for this
measurement only: commit "Pooling in the GEMM launch" + 1) -- ms per step (G = 4 x 2 lanes) and the six GEMM
launches, one subprocess per library, three interleaved rounds; the split tests run against each variant first."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = open(os.path.join(ROOT, "profiles", "experiments", "ilv_threshold.py")).read().split("WORKER = r'''")[1].split("'''")[0]
libs = {"shipped": None, "seq4 up-front": os.path.join(ROOT, "scratch", "libs", "libbbb_seq4_1.so"),
        "seq4 interleaved": os.path.join(ROOT, "scratch", "libs", "libbbb_seq4_2.so")}
for tag, lib in libs.items():
    if lib:
        env = dict(os.environ, BBB_HIP_LIB=lib)
        p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_splitk.py"), "-q", "-x"], capture_output=True, text=True, env=env, timeout=900)
        print(json.dumps({"lib": tag, "split_tests": p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-200:]}), flush=True)
for rnd in range(3):
    for tag, lib in libs.items():
        env = dict(os.environ)
        if lib:
            env["BBB_HIP_LIB"] = lib
        p = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        print(json.dumps({"lib": tag, **(json.loads(line[0][7:]) if line else {"error": p.stderr[-300:]})}), flush=True)
