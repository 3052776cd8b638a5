import os
for f in ('/sys/fs/cgroup/cpu.max','/sys/fs/cgroup/cpu.stat','/sys/fs/cgroup/cpu/cpu.cfs_quota_us','/sys/fs/cgroup/cpu/cpu.cfs_period_us','/sys/fs/cgroup/cpu/cpu.stat'):
    try: print(f, open(f).read().replace('\n',' | '))
    except Exception as e: print(f, 'n/a')
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
