import os, sys, shutil
lib = sys.argv[1]
shutil.copy(lib, "tzddpc_amd/lib/libtzddpc_hip.so")
os.execv(sys.executable, [sys.executable, "tools/gpu_prof.py"] + sys.argv[2:])
