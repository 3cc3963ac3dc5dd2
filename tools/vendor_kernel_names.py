import torch
for M, N, K in ((14592, 3072, 768), (5120, 3072, 768), (9472, 768, 2048), (14592, 768, 3072)):
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); B = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(A, B.t())
    A2 = torch.randn(M, N, device="cuda", dtype=torch.bfloat16); B2 = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(A2, B2)
torch.cuda.synchronize()
