#!/bin/bash
# A/B of the grouped weight-gradient tile per stack on the ViT-B step
python3 tools/step_only.py
for t in 64064 128128 9128128; do echo "DEC=$t"; SKYEMB_WGRAD_TILE_DEC=$t python3 tools/step_only.py; done
for t in 64064 128064 9128128; do echo "ENC=$t"; SKYEMB_WGRAD_TILE_ENC=$t python3 tools/step_only.py; done
