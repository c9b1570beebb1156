"""Robot policies: SARL attention value network + 81-action look-ahead, SGAN world model."""
