"""Oracle package: CPU restatement of the reference hot path. TEST INFRASTRUCTURE ONLY (see gt_oracle.py)."""
