import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from qml_essentials_amd import operations as op
from qml_essentials_amd.script import Script
from qml_essentials_amd.model import Model
from qml_essentials_amd import jaqsi as js

def circ(th, x):
    op.H(wires=0); op.RX(th[0], wires=0); op.RY(th[1], wires=1); op.CX(wires=[0,1])
    op.CRX(th[2], wires=[1,2]); op.Rot(th[3], th[4], th[5], wires=2); op.CRZ(th[6]*x, wires=[2,0])
    op.ControlledPhaseShift(th[7], wires=[0,1]); op.RXX(th[8], wires=[1,2]); op.RZZ(th[9], wires=[0,2])
    op.RYY(th[10], wires=[0,1]); op.RZX(th[11], wires=[2,1]); op.CRY(th[12], wires=[0,2]); op.S(wires=1); op.RZ(th[13], wires=1)
    op.RY(th[14], wires=0)

s = Script(circ, n_qubits=3)
rng = np.random.default_rng(0)
th = rng.uniform(0, 6.28, 15); x = 0.7
obs = [op.PauliZ(wires=0, record=False), op.PauliZ(wires=2, record=False), js.build_parity_observable([0,1,2])]
(jac, jx) = s.gradient(obs, args=(th, np.array(x)), argnums=(0,1))
w = np.array([0.3, -1.1, 0.8])
(g, gx) = s.vjp(obs, w, args=(th, np.array(x)), argnums=(0,1))
print("max diff params", np.abs(g - w @ jac).max(), "x", abs(gx - w @ jx))
print(g[:5], (w@jac)[:5])
# batched
TH = rng.uniform(0, 6.28, (5, 15)); X = rng.uniform(0,1,5)
(jac, jx) = s.gradient(obs, args=(TH, X), in_axes=(0,0), argnums=(0,1))
W = rng.normal(size=(5,3))
(g, gx) = s.vjp(obs, W, args=(TH, X), in_axes=(0,0), argnums=(0,1))
print("batched max diff", np.abs(g - np.einsum('bk,bkp->bp', W, jac)).max(), np.abs(gx - np.einsum('bk,bk->b', W, jx)).max())
