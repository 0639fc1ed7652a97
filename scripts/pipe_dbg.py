import sys, time, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, ".")
from imagescry_amd import EmbeddingBank, EmbedSearchPipeline, ImageBatch, ViTB16Embedder
import bench
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
shard = bench.make_shard(0, n, 768, dev)
bank = EmbeddingBank(shard, dtype=torch.float16, normalize=False)
del shard
model = ViTB16Embedder(seed=0).to(dev)
images = torch.randint(0, 256, (512, 3, 224, 224), dtype=torch.uint8, generator=torch.Generator().manual_seed(1234)).to(dev)
batch = ImageBatch(indices=torch.arange(512, device=dev), images=images)
emb = model.predict_step(batch)
q = emb.get_flat_vectors().half()
print("query norms", q.float().norm(dim=1)[:4].tolist(), "pairwise cos min", float((torch.nn.functional.normalize(q.float(), dim=1) @ torch.nn.functional.normalize(q.float(), dim=1).T).min()))
for name, qq in (("vit embeddings", q), ("random", torch.randn(512, 768, device=dev).half())):
    for _ in range(2): bank.search(qq, 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): s, i = bank.search(qq, 10)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(name, f"{dt*1e3:.2f} ms", "status", bank.last_status.tolist(), "top scores", s[0, :3].tolist(), s[0, -1].item())
