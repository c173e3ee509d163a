import sys; sys.path.insert(0, "/root/repo")
import torch, mplan2vdl_amd as m
from mplan2vdl_amd import datagen
e = m.Engine(0)
keep = datagen.register_q3_columns(e, 1500000)
p = e.parse(open("/root/repo/tests/golden/q3.vdl").read())
q6 = None
for it in range(300):
    p.execute()
    if it in (5, 50, 150, 299):
        free, total = torch.cuda.mem_get_info()
        print("after %3d runs: %.1f MB of device memory in use" % (it + 1, (total - free) / 1e6), flush=True)
e.close()
free, total = torch.cuda.mem_get_info()
print("after close: %.1f MB in use" % ((total - free) / 1e6))
