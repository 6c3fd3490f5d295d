import torch
torch.cuda.set_device(0); torch.cuda.synchronize()
exec(open("profiles/scripts/two_streams.py").read())
