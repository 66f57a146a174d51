"""MI355X-native EM-Dirichlet / Hard EM-Dirichlet transductive-inference engine (host side)."""
