import torch
d = torch.device("cuda", 0)
torch.cuda.set_device(0)
print("default", torch.cuda.default_stream(d).cuda_stream, "current", torch.cuda.current_stream(d).cuda_stream)
for i in range(3):
    e = torch.cuda.ExternalStream(0, device=d)
    print("ExternalStream(0) ->", e.cuda_stream, e)
