import sys, subprocess
cases = [(8,6,4,2,40,1.0,0),(8,40,22,2,120,1.0,0),(8,40,22,2,120,.15,0),(8,40,22,2,120,.15,1),(8,40,22,2,120,0.0,1),(8,9,5,1,64,.3,2)]
for c in cases:
    code = "import sys; sys.path.insert(0,'tests'); import test_gpu_h264_decoder as T; T._run_picture(%d,%d,%d,%d,%d,%r,%d,seed=5); print('OK')" % c
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, env=dict(__import__('os').environ, AMD_LOG_LEVEL="1"))
    tail = [l for l in (r.stdout + r.stderr).splitlines() if not l.startswith("  File") and "Extension modules" not in l][-6:]
    print(c, "rc", r.returncode, " | ".join(t[:200] for t in tail), flush=True)
