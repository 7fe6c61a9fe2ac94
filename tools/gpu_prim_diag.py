import torch, subprocess
print(subprocess.run("lscpu | grep -E 'Model name|Flags' | cut -c1-200", shell=True, capture_output=True, text=True).stdout)
print(torch.__config__.show().split("CPU capability")[1][:40])
dev = torch.device("cuda:0")
torch.manual_seed(0)
a = torch.rand(1 << 20) + 0.1
exact = torch.sqrt(a.double()).float()
print("cpu torch.sqrt vs exact :", float((torch.sqrt(a) != exact).float().mean()))
print("gpu torch.sqrt vs exact :", float((torch.sqrt(a.to(dev)).cpu() != exact).float().mean()))
print("gpu f64 sqrt->f32 vs exact:", float((torch.sqrt(a.to(dev).double()).float().cpu() != exact).float().mean()))
for s in (5.0, 63.0):
    td = (a.double() / s).float()
    print("div", s, "cpu vs true:", float(((a / s) != td).float().mean()), " gpu vs true:", float(((a.to(dev) / s).cpu() != td).float().mean()))
b = torch.rand(1 << 20) + 0.1
td = (a.double() / b.double()).float()
print("tensor div cpu vs true:", float(((a / b) != td).float().mean()), " gpu:", float(((a.to(dev) / b.to(dev)).cpu() != td).float().mean()))
