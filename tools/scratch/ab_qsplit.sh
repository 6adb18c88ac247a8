#!/bin/bash
mk() { echo "[setattr(c,'split_cap',$1) for mm in model.recognizer.q_modules() for c in mm.modules() if hasattr(c,'split_cap')]"; }
python tools/ab_step.py "$(mk 16)" "$(mk 8)" "$(mk 4)" --rounds 5 2>&1 | grep -v amdgpu | sed 's/^.*modules() if hasattr(c,.split_cap.)\]//' 
