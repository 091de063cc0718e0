"""CPU oracle of the reference's Tacotron decoder hot path.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py import this package; the product package
(`torch-tts_amd/`, imported as `torch_tts_amd`) never does.
"""
