VARIANTS="" BPBS="0,1,2,3,4,6,8" ALLOCS=3 timeout 600 python tools/placement_ab.py 2>&1 | grep alloc
