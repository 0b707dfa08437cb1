T="tests/test_gpu_policy.py::test_quad_rule_within_3_percent_of_the_other_path[2048-2048]"
for pr in "" 0 1; do
  echo "== ISING_QUAD_PRIO=$pr after test_gpu_ballot"
  if [ -n "$pr" ]; then export ISING_QUAD_PRIO=$pr; else unset ISING_QUAD_PRIO; fi
  python -m pytest tests/test_gpu_ballot.py "$T" -q -s 2>&1 | grep -E "the library \(|passed|failed" | grep -v print
done
unset ISING_QUAD_PRIO
for f in $(python - <<'PY'
import re
src=open("tests/test_gpu_ballot.py").read()
print(" ".join(re.findall(r"^def (test_\w+)", src, re.M)))
PY
); do
  echo "== after $f"
  python -m pytest "tests/test_gpu_ballot.py::$f" "$T" -q -s 2>&1 | grep -E "the library \(|passed|failed" | grep -v print
done
