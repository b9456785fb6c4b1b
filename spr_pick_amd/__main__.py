from .cli import start

if __name__ == "__main__":
    import torch.distributed as dist
    start()
    # success path only: a rank that failed must exit non-zero NOW — tearing the RCCL communicator down waits for the
    # peers' outstanding collectives (until the watchdog timeout) and could mask the original exception
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
