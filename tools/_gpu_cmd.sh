./tools/micro/store_patterns 2>&1 | grep -E "^[5789] |^[34] " 
