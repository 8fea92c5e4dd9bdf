#!/usr/bin/env python3
"""Generate tests/golden/topologies.json from the *real* reference module.

The only hot-path-adjacent reference modules that import without JAX are
``qml_essentials.topologies`` and ``qml_essentials.tape``.
This script imports ``/root/reference/qml_essentials/topologies.py`` and records
the ``[control, target]`` pair lists for every entangling block of every ansatz
(call signatures transcribed from ``qml_essentials/ansaetze.py:437-756``) for
n_qubits = 2..8.  The JSON is DATA (inputs -> expected outputs); no reference
source text is stored.

Run (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_topology_fixtures.py
"""
import json
import os
import sys

sys.path.insert(0, "/root/reference")
from qml_essentials.topologies import Topology  # noqa: E402

# (fixture key, topology name, kwargs) -- kwargs as written at the cited line.
CALLS = [
    ("GHZ.1", "stairs", dict(reverse=True)),  # ansaetze.py:420
    ("Circuit_2.2", "stairs", dict()),  # :452-455
    ("Circuit_3.2", "stairs", dict()),  # :464
    ("Circuit_4.2", "stairs", dict()),  # :473
    ("Circuit_5.2", "all_to_all", dict()),  # :482
    ("Circuit_6.2", "all_to_all", dict()),  # :493
    ("Circuit_7.2", "bricks", dict()),  # :505-508
    ("Circuit_7.5", "bricks", dict(offset=1)),  # :511-515
    ("Circuit_8.2", "bricks", dict()),  # :524-527
    ("Circuit_8.5", "bricks", dict(offset=1)),  # :530-534
    ("Circuit_9.1", "stairs", dict()),  # :542
    ("Circuit_10.1", "stairs", dict(offset=-1, wrap=True)),  # :551
    ("Circuit_13.1", "stairs", dict(wrap=True, reverse=True, mirror=False)),  # :560-566
    ("Circuit_13.3", "stairs",
     dict(reverse=False, mirror=False, offset="n-1", span=3, wrap=True)),  # :568-576
    ("Circuit_14.1", "stairs", dict(wrap=True, reverse=True, mirror=False)),  # :584-590
    ("Circuit_14.3", "stairs",
     dict(reverse=False, mirror=False, offset="n-1", span=3, wrap=True)),  # :592-600
    ("Circuit_15.1", "stairs", dict(wrap=True, reverse=True, mirror=False)),  # :608-614
    ("Circuit_15.3", "stairs",
     dict(reverse=False, mirror=False, offset="n-1", span=3, wrap=True)),  # :616-624
    ("Circuit_16.2", "bricks", dict()),  # :633-636
    ("Circuit_16.3", "bricks", dict(offset=1)),  # :637-641
    ("Circuit_17.2", "bricks", dict()),  # :650-653
    ("Circuit_17.3", "bricks", dict(offset=1)),  # :654-658
    ("Circuit_18.2", "stairs", dict(wrap=True, mirror=False)),  # :667-672
    ("Circuit_19.2", "stairs", dict(wrap=True, mirror=False)),  # :681-686
    ("Circuit_20.1", "stairs", dict(wrap=True, reverse=True, mirror=False)),  # :694-700
    ("Circuit_20.3", "stairs",
     dict(reverse=False, offset="n-2", span=1, wrap=True)),  # :702-709
    ("Hardware_Efficient.3", "bricks", dict(mirror=False)),  # :723-727
    ("Hardware_Efficient.4", "bricks",
     dict(offset=-1, modulo=True, wrap=True, mirror=False)),  # :728-735
    ("Strongly_Entangling.1", "stairs",
     dict(wrap=True, reverse=False, mirror=False)),  # :743-749
    ("Strongly_Entangling.3", "stairs",
     dict(reverse=False, span="n//2", wrap=True, mirror=False)),  # :751-758
]

_SYMBOLIC = {
    "n-1": lambda n: n - 1,
    "n-2": lambda n: n - 2,
    "n//2": lambda n: n // 2,
}


def main() -> None:
    out = {"_source": "qml_essentials.topologies.Topology (reference v0.2.2)",
           "n_qubits": list(range(2, 9)), "calls": {}}
    for key, topo, kwargs in CALLS:
        real_kwargs = {
            k: (_SYMBOLIC[v] if isinstance(v, str) else v) for k, v in kwargs.items()
        }
        per_n = {}
        for n in range(2, 9):
            pairs = getattr(Topology, topo)(n_qubits=n, **real_kwargs)
            per_n[str(n)] = [[int(a), int(b)] for a, b in pairs]
        out["calls"][key] = {"topology": topo, "kwargs": kwargs, "pairs": per_n}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "topologies.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", path, len(out["calls"]), "calls")


if __name__ == "__main__":
    main()
