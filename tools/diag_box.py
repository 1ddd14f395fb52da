import os, time, subprocess
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max",):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, e)
print(subprocess.run("nproc; lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket'; free -g | head -2", shell=True, capture_output=True, text=True).stdout)
