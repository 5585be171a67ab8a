OUT=gpurun_out/r03a; mkdir -p $OUT
python - <<'PY' > $OUT/c5_lanes.jsonl 2>&1
import json, os, subprocess, sys
code = open("tools/c5_probe.py").read()
for lanes in ("64", "32", "16", "8"):
    env = dict(os.environ, MISPEC_SHIFT_LANES=lanes)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    print(r.stdout.strip() or json.dumps({"lanes": lanes, "error": r.stderr[-400:]}), flush=True)
PY
cat $OUT/c5_lanes.jsonl
