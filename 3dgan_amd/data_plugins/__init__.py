"""Dataset plugins (the reference's `hem/data/*.py`): one class per dataset whose first base is `DataPlugin`, found by
3dgan_amd/plugins.py.  Each serves per-replica batches already resident in HBM (`get_source`)."""
