"""Import-only stand-in for torch_geometric (absent here); see torch_scatter stub."""
