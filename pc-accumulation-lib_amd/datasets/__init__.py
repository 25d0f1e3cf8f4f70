"""Dataset helpers of the drop-in root.  A real package (with __init__) on purpose: the reference's
``datasets`` is a namespace directory and loses the import race against an installed HuggingFace
``datasets`` distribution."""
