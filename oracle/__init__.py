"""oracle — CPU restatement of RAGRoute's retrieval hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
ragroute_amd/ (the product) never does.
"""
