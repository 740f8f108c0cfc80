"""MI355X-native ASE / AMP training update.

Hardware queues.  The update engine runs the three network branches of an optimisation step (policy | critic |
discriminator) on three HIP streams.  How many streams the runtime lets execute side by side is its hardware-queue
count, ``GPU_MAX_HW_QUEUES``, which libamdhip64 reads ONCE when it initialises (the first HIP call of the process, not
``import torch``).  Measured on MI355X with the launch programs of this package: 75.9 ms per update with 4 queues,
78.5 ms with 3 (the null stream's work shares a queue with one branch), 80.5 ms with 6.  Importing this package sets the
variable to 4 unless the user set it; if HIP is already initialised by then the setting cannot take effect any more and
``hw_queue_note`` says so.  Nothing depends on the value for correctness: with fewer queues than streams the branches
simply serialise (the fork / join points are events, replayed by the library itself - no hipGraph involved).
"""
import os
import sys

HW_QUEUES_DEFAULT = '4'
hw_queue_note = None


def _configure_hw_queues():
    global hw_queue_note
    if 'GPU_MAX_HW_QUEUES' in os.environ:
        hw_queue_note = f"GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']} (set by the user)"
        return
    t = sys.modules.get('torch')
    if t is not None and getattr(t, 'cuda', None) is not None and t.cuda.is_initialized():
        hw_queue_note = ("HIP was initialised before `import ase_amd`: GPU_MAX_HW_QUEUES keeps the runtime default; import "
                         "ase_amd (or export GPU_MAX_HW_QUEUES=4) before the first GPU call for the measured stream overlap")
        return
    os.environ['GPU_MAX_HW_QUEUES'] = HW_QUEUES_DEFAULT
    hw_queue_note = f"GPU_MAX_HW_QUEUES={HW_QUEUES_DEFAULT} (set by ase_amd)"


_configure_hw_queues()
