#!/bin/bash
# Same-box A/B of two builds of libwsae_hip.so: alternate A, B, A, B ... (bench.py --profile-all for the per-kernel
# event times, then a plain run for the step), print one line per run.
#   profiles/tools/ab.sh build_ab/libwsae_hip_b.so [rounds]        (A = the in-tree library)
set -e
B_LIB=$1; ROUNDS=${2:-2}
A_LIB=${A_LIB:-whisper-sae_amd/whisper_sae/libwsae_hip.so}
mkdir -p gpurun_out
python3 - "$A_LIB" "$B_LIB" "$ROUNDS" <<'PY'
import json, os, subprocess, sys
a, b, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
for r in range(rounds):
    for tag, lib in (("A", a), ("B", b)):
        env = dict(os.environ, WSAE_LIB=os.path.abspath(lib))
        o = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "100", "--warmup", "100", "--profile-all"],
                           env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(tag, "FAILED", o.stderr[-800:]); continue
        j = json.loads(line[-1])
        k = {n.split("<")[0].replace("_kernel", ""): round(v * 1e3, 1) for n, v in j["kernel_ms_per_step"].items()}
        o2 = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "100", "--warmup", "100"],
                            env=env, capture_output=True, text=True)
        j2 = json.loads([l for l in o2.stdout.splitlines() if l.startswith("{")][-1])
        print(tag, "step_us", round(j2["ms_per_step"] * 1e3, 1), "loss", j2["final_loss"], k, flush=True)
PY
