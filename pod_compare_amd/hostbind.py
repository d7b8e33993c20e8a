"""Host threads of a rank next to its GPU.

On an 8-GPU node every rank runs its enqueue thread, the `DATALOADER.NUM_WORKERS` loader threads of apply_net.Prefetched (AN:83-84's data loader)
and torch's intra-op threads (AN:33-40 sets 32): 8 ranks x (1 + workers) threads that the kernel scheduler is otherwise free to put on the
other socket -- pinned staging buffers and the doorbell writes of ~200 launches per image then cross the socket link.  `bind_rank_to_gpu_numa`
restricts the calling process (every thread started afterwards inherits it) to the CPUs of the NUMA node its GPU hangs off, read from sysfs
(/sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node -> /sys/devices/system/node/node<k>/cpulist).  Nothing is guessed: a box without the
sysfs entries (a container, a single-node host: numa_node = -1) is left alone and the returned record says so; bench.py / apply_net print the
record (`config.host_binding`)."""
import os
from typing import Dict, List, Optional


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def pci_address(device_index: int) -> Optional[str]:
    import torch
    p = torch.cuda.get_device_properties(device_index)
    if not all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        return None
    return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)


def numa_node_of(pci: Optional[str], sysfs: str = "/sys") -> Optional[int]:
    if pci is None:
        return None
    try:
        node = int(open(os.path.join(sysfs, "bus/pci/devices", pci, "numa_node")).read())
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


def cpus_of_node(node: int, sysfs: str = "/sys") -> List[int]:
    try:
        return _parse_cpulist(open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)).read())
    except OSError:
        return []


def bind_rank_to_gpu_numa(device_index: int, enable: Optional[bool] = None, sysfs: str = "/sys") -> Dict:
    """Restricts this process to the CPUs of `device_index`'s NUMA node.  enable: None = POD_BIND_NUMA (default on), False = only report."""
    if enable is None:
        enable = os.environ.get("POD_BIND_NUMA", "1") != "0"
    pci = pci_address(device_index)
    node = numa_node_of(pci, sysfs)
    rec = {"device": device_index, "pci": pci, "numa_node": node, "bound": False, "cpus": None}
    if node is None:
        rec["why_not"] = "no NUMA node for the device in sysfs (container / single-node host)"
        return rec
    cpus = [c for c in cpus_of_node(node, sysfs) if c in os.sched_getaffinity(0)]
    if not cpus:
        rec["why_not"] = "the node's CPUs are outside this process's affinity mask"
        return rec
    rec["cpus"] = "%d CPUs: %d-%d" % (len(cpus), cpus[0], cpus[-1])
    if enable:
        os.sched_setaffinity(0, cpus)
        rec["bound"] = True
    return rec
