set -x
timeout 600 python examples/train_s2s_pretrain.py --epochs 2 --clips 12 --batch 4 --max-len 60 --out /tmp/slm.pt 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8
timeout 600 python examples/finetune_s2s_pretrain.py --epochs 1 --clips 12 --batch 4 --max-len 60 --out /tmp/ft.pt 2>&1 | tail -4
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6
