import sys, json
sys.path.insert(0, '.')
from tests import model_checks as C
for name in ("C", "A"):
    for r in C.hip_full_model_checks(name):
        print(json.dumps({k: (v if not hasattr(v, 'item') else float(v)) for k, v in r.items()}))
for name in ("A", "C"):
    for r in C.hip_grad_checks(name):
        print(json.dumps(r))
