import torch
for p in (-2,-1,0,1,2):
    try:
        s = torch.cuda.Stream(priority=p); print(p, "ok", s.priority)
    except Exception as e:
        print(p, "err", str(e)[:80])
