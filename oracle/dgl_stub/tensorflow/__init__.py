"""Empty stand-in so `import tensorflow as tf` in the reference's AmazonDataset
module (imported transitively by utils.py:9) succeeds.  Nothing is called."""
