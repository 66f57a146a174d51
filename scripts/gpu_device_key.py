import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "transductive-clip_amd"))
import torch
from tclip_amd import sharding
p = torch.cuda.get_device_properties(0)
print("uuid", getattr(p, "uuid", None), "pci", [getattr(p, n, None) for n in ("pci_domain_id", "pci_bus_id", "pci_device_id")], "name", p.name)
print("device_key", sharding._device_key(0), "node_key", sharding._node_key())
