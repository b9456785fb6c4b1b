from .cli import start

if __name__ == "__main__":
    try:
        start()
    finally:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()     # RCCL: tear the communicator down before the interpreter exits
