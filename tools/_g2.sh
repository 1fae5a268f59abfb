for rep in 1 2 3; do
  echo "step inherit:    $(python tools/prof_step.py bf16x3 30 2>/dev/null | grep ms/step)"
  echo "step full patch: $(CRN_BF3_WG_INHERIT=0 python tools/prof_step.py bf16x3 30 2>/dev/null | grep ms/step)"
done
