"""What the bench host gives a process: CPU affinity, SMT siblings, the cgroup CPU quota (MI355X boxes of this pool: 256 CPUs visible,
siblings c / c + 128, quota 16 CPUs -- the launcher and the draw thread float and can land on two siblings of one core)."""
import os
aff=sorted(os.sched_getaffinity(0)); print('affinity',len(aff),aff)
for c in aff[:20]:
    try: print(c, open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list'%c).read().strip(), open('/sys/devices/system/cpu/cpu%d/topology/physical_package_id'%c).read().strip())
    except Exception as e: print(c, e)
print(open('/sys/fs/cgroup/cpu.max').read() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'no cpu.max')
