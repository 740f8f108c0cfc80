"""MI355X-native ASE / AMP training update.

Hardware queues.  The update engine runs the three network branches of an optimisation step (policy | critic |
discriminator) on three HIP streams - four with the gradient penalty's value path of the gp_f32 modes, and the agents run the
update on a high-priority stream of their own.  How many streams the runtime lets execute side by side is its hardware-queue
count, ``GPU_MAX_HW_QUEUES``, which libamdhip64 reads ONCE when it initialises (the first HIP call of the process, not
``import torch``).  Measured on MI355X with the launch programs of this package: 75.9 ms per update with 4 queues,
78.5 ms with 3 (the null stream's work shares a queue with one branch), 80.5 ms with 6.

Importing this package changes NOTHING in the process.  A launcher that wants the measured overlap calls
``ase_amd.configure()`` before its first GPU call (bench.py does); the call reports whether the setting could still take
effect (``ase_amd.hw_queue_note``) and the bench line carries that note.  Nothing depends on the value for correctness: with
fewer queues than streams the branches simply serialise (the fork / join points are events, replayed by the library itself -
no hipGraph involved).
"""
import os
import sys

HW_QUEUES_DEFAULT = '4'
hw_queue_note = 'ase_amd.configure() was not called: GPU_MAX_HW_QUEUES keeps the runtime default'
hw_queues_applied = False


def configure(hw_queues=HW_QUEUES_DEFAULT, cpu_threads=None):
    """Process-level runtime settings for the measured stream overlap; call BEFORE the first HIP call of the process.
    Returns True if the setting is in effect (set here, or already exported by the user), False if HIP was initialised
    earlier (then only a warning note is left).

    cpu_threads: torch's intra-op CPU thread count (torch.set_num_threads).  The update keeps the host one step of work
    ahead of the GPU at most; torch sizes its OpenMP pool by the machine (128 threads on a 256-core host) even inside a
    container with a 16-core CPU quota, every small CPU tensor operation leaves those workers spinning, the cgroup throttles the
    whole process for the rest of its 100 ms period, and the GPU idles: measured 2 updates in 20 at 1.2-2.3 x their 65 ms, none
    in 120 with one thread (profiles/r03_update_time_spikes_threads.log).  bench.py passes 1; None leaves torch alone."""
    global hw_queue_note, hw_queues_applied
    if cpu_threads is not None:
        import torch
        torch.set_num_threads(max(1, int(cpu_threads)))
    if 'GPU_MAX_HW_QUEUES' in os.environ:
        hw_queue_note = f"GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']} (set by the user)"
        hw_queues_applied = True
        return True
    t = sys.modules.get('torch')
    if t is not None and getattr(t, 'cuda', None) is not None and t.cuda.is_initialized():
        hw_queue_note = ("WARNING: HIP was initialised before ase_amd.configure(): GPU_MAX_HW_QUEUES keeps the runtime default "
                         "(call configure() or export GPU_MAX_HW_QUEUES=4 before the first GPU call for the measured stream overlap)")
        hw_queues_applied = False
        return False
    os.environ['GPU_MAX_HW_QUEUES'] = str(hw_queues)
    hw_queue_note = f"GPU_MAX_HW_QUEUES={hw_queues} (set by ase_amd.configure())"
    hw_queues_applied = True
    return True
