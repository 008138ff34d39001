"""Test oracle: CPU restatement of the reference's hot path ("Oracle").

TEST INFRASTRUCTURE ONLY.  Nothing under scaling_retriever_amd/ imports this
package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
"""
