import sys; sys.path.insert(0,'.')
from gumbi_amd import engine
print(engine.mfma_f64_sustained(0, 0.5))
