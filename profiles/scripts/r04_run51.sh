# whole GPU suite on the committed state
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -5
