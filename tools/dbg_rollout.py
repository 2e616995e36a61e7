import subprocess, sys, os
ROOT = "/root/repo"
ids = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "--co", "-k", "test_rollout_equals_steps"], capture_output=True, text=True, cwd=ROOT).stdout.splitlines()
ids = [l for l in ids if "::" in l]
print(len(ids), "ids")
for i in ids:
    try:
        r = subprocess.run([sys.executable, "-m", "pytest", i, "-x", "-q"], capture_output=True, text=True, cwd=ROOT, timeout=120)
        tail = [l for l in r.stdout.splitlines() if l.strip()][-1:] 
        print(i.split("::")[-1], "rc", r.returncode, tail, flush=True)
        if r.returncode not in (0,):
            print("\n".join(r.stdout.splitlines()[-25:]))
            print(r.stderr[-600:])
    except subprocess.TimeoutExpired:
        print(i, "TIMEOUT", flush=True)
