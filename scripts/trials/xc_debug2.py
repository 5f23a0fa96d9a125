import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import kernel_cases as KC
for kw in (dict(n=8, tokens=4096, clip=8, front=True, seed=2), dict(n=16, tokens=4096, clip=8, front=True, seed=3),
           dict(n=3, tokens=1024, clip=2, front=True, bias=False, ln=False, seed=4, lk=60), dict(n=8, tokens=4096, clip=8, front=True, seed=12),
           dict(n=8, tokens=4096, clip=8, front=True, seed=13, bias=False)):
    for rep in range(2):
        print(kw, KC.case_xattn_chain("cuda", exact=False, **kw), flush=True)
